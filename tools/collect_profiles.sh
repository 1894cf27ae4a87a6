#!/bin/bash
# usage: tools/collect_profiles.sh TAG
# Runs on the GPU box: the default bench (JSON line), a rocprofv3 kernel trace of the same command, the
# FETCH_SIZE / WRITE_SIZE passes and two SQ-counter passes over `bench.py --pmc-target`; copies the
# summaries into profiles/ under TAG.
T=${1:-r1}
rm -rf gpurun_out/$T
/usr/local/graft/bin/gpurun --timeout 1200 -- "mkdir -p gpurun_out/$T/kt gpurun_out/$T/f gpurun_out/$T/w gpurun_out/$T/s1 gpurun_out/$T/s2; R=\$GRAFT_REPO_ROOT; O=\$R/gpurun_out/$T; cd /tmp && export TMPDIR=/tmp && \
timeout -k 10 400 python3 \$R/bench.py > \$O/bench.json 2> \$O/bench.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d \$O/kt -- python3 \$R/bench.py --no-cpu-baseline --no-inference --no-networks > \$O/kt/run.log 2>&1 && \
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d \$O/f -- python3 \$R/bench.py --pmc-target > \$O/f/run.log 2>&1 && \
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d \$O/w -- python3 \$R/bench.py --pmc-target > \$O/w/run.log 2>&1 && \
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d \$O/s1 -- python3 \$R/bench.py --pmc-target > \$O/s1/run.log 2>&1 && \
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d \$O/s2 -- python3 \$R/bench.py --pmc-target > \$O/s2/run.log 2>&1; echo exit=\$?" 2>&1 | tail -2
O=gpurun_out/$T
mkdir -p profiles
cp $O/bench.json profiles/${T}_bench.json
cp $O/kt/vm/*_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
python tools/step_breakdown.py $O/kt/vm/*_kernel_trace.csv 20 > profiles/${T}_bench_step_breakdown.txt
python tools/pmc_summary.py $O/f $O/w profiles/${T}_pmc_traffic.json > profiles/${T}_pmc_traffic.txt
cp profiles/${T}_pmc_traffic.json profiles/pmc_traffic_latest.json
cp $O/f/vm/*_counter_collection.csv profiles/${T}_pmc_fetch_size.csv
cp $O/w/vm/*_counter_collection.csv profiles/${T}_pmc_write_size.csv
python tools/pmc_table.py $O/s1 $O/s2 > profiles/${T}_pmc_sq_counters.txt
python tools/show_bench.py < $O/bench.json
