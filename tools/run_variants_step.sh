#!/bin/bash
# usage (on the GPU box): tools/run_variants_step.sh NAME...  -> ms per bench step (100 steps, twice) with each variant library, then with the real one
L=visiontransformer_amd/libdswin_hip.so
cp $L /tmp/_orig.so
one() { for i in 1 2; do timeout -k 10 200 python bench.py --no-kernels --no-inference --no-networks --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('  ms_per_step', round(d['ms_per_step'],4))"; done; }
for n in "$@"; do echo "== variant $n"; cp visiontransformer_amd/variants/libdswin_$n.so $L; one; done
cp /tmp/_orig.so $L; echo "== real"; one
