#!/bin/bash
# usage: tools/pmc_sq.sh NAME -> two SQ-counter passes over `bench.py --pmc-target`, per-kernel table for the attention kernels
N=${1:-sq}
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/${N}_1 gpurun_out/${N}_2; R=\$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp && timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d \$R/gpurun_out/${N}_1 -- python3 \$R/bench.py --pmc-target > \$R/gpurun_out/${N}_1/run.log 2>&1 && timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d \$R/gpurun_out/${N}_2 -- python3 \$R/bench.py --pmc-target > \$R/gpurun_out/${N}_2/run.log 2>&1; echo exit=\$?" 2>&1 | tail -1
python tools/pmc_table.py gpurun_out/${N}_1 gpurun_out/${N}_2
