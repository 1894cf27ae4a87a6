#!/bin/bash
# on the GPU box: tools/pmc_w16.sh OUT  -> SQ-counter passes over the 16x16-window attention microbenchmark (WMSA shape, padded rows)
O=$GRAFT_REPO_ROOT/gpurun_out/$1; R=$GRAFT_REPO_ROOT
mkdir -p $O/s1 $O/s2
cd /tmp && export TMPDIR=/tmp
export W16_SHAPE=wmsa W16_PITCH=padded W16_ITERS=3 W16_ROUNDS=1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/s1 -- python3 $R/tools/attn_w16_bench.py > $O/s1/run.log 2>&1 || exit 4
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/s2 -- python3 $R/tools/attn_w16_bench.py > $O/s2/run.log 2>&1 || exit 5
python3 $R/tools/pmc_table.py $O/s1 $O/s2 > $O/table.txt
