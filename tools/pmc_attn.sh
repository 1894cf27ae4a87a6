#!/bin/bash
# on the GPU box: tools/pmc_attn.sh OUT  -> kernel trace + two SQ-counter passes over the attention microbenchmark
O=$GRAFT_REPO_ROOT/gpurun_out/$1; R=$GRAFT_REPO_ROOT
mkdir -p $O/kt $O/s1 $O/s2 $O/s3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/attn_microbench.py > $O/kt/run.log 2>&1 || exit 3
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/s1 -- python3 $R/tools/attn_microbench.py > $O/s1/run.log 2>&1 || exit 4
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM_WR SQ_ACTIVE_INST_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/s2 -- python3 $R/tools/attn_microbench.py > $O/s2/run.log 2>&1 || exit 5
timeout -s KILL 90 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/s3 -- python3 $R/tools/attn_microbench.py > $O/s3/run.log 2>&1 || exit 6
