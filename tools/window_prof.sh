#!/bin/bash
# usage: tools/window_prof.sh NAME [args of window_bench.py] -> rocprofv3 per-kernel averages of tools/window_bench.py on the GPU box
N=${1:-wb}; shift
/usr/local/graft/bin/gpurun --timeout 600 -- "mkdir -p gpurun_out/$N; cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/$N -- python3 \$GRAFT_REPO_ROOT/tools/window_bench.py $* > \$GRAFT_REPO_ROOT/gpurun_out/$N/run.log 2>&1; echo exit=\$?" 2>&1 | grep -E "exit=|charged"
python - <<PY
import csv, glob
f = sorted(glob.glob('gpurun_out/$N/*/*_kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if 'dsw::' in r['Name']:
        print("%-100s n=%s avg=%.1f us" % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3))
PY
