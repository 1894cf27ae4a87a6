#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG ..."  -> visiontransformer_amd/variants/libdswin_NAME.so (experiments only)
set -e
N=$1; shift
mkdir -p build/var_$N visiontransformer_amd/variants
for f in visiontransformer_amd/csrc/*.hip; do
  o=build/var_$N/$(basename $f).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=off -Iinclude "$@" -c $f -o $o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o visiontransformer_amd/variants/libdswin_$N.so build/var_$N/*.o
echo built $N
