#!/bin/bash
# on the GPU box: tools/pmc_variants.sh NAME...  -> LDS counters of the attention backward per variant library
L=visiontransformer_amd/libdswin_hip.so; cp $L /tmp/_orig.so; R=$GRAFT_REPO_ROOT
for n in "$@"; do
  cp visiontransformer_amd/variants/libdswin_$n.so $L
  mkdir -p $R/gpurun_out/pv_$n
  (cd /tmp && TMPDIR=/tmp timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pv_$n -- python3 $R/tools/attn_microbench.py > $R/gpurun_out/pv_$n/run.log 2>&1) || break
done
cp /tmp/_orig.so $L
