#!/bin/bash
# usage (on the GPU box): tools/run_variants.sh SCRIPT NAME...  -> runs SCRIPT once per variant library, then with the real one
S=$1; shift
L=visiontransformer_amd/libdswin_hip.so
cp $L /tmp/_orig.so
for n in "$@"; do
  echo "== variant $n"; cp visiontransformer_amd/variants/libdswin_$n.so $L; timeout -k 10 200 python $S || true
done
cp /tmp/_orig.so $L; echo "== real"; timeout -k 10 200 python $S
