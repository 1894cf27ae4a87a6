#!/bin/bash
# usage: tools/prof_step.sh NAME  -> rocprofv3 kernel trace of a 10-step bench run on the GPU box, breakdown printed
N=${1:-p}
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/$N; cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/$N -- python3 \$GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernels --no-inference --no-networks > \$GRAFT_REPO_ROOT/gpurun_out/$N/run.log 2>&1; echo exit=\$?" 2>&1 | tail -2
python tools/step_breakdown.py gpurun_out/$N/vm/*_kernel_trace.csv 10 | cut -c1-130
