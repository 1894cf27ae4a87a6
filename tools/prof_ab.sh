#!/bin/bash
# on the GPU box: tools/prof_ab.sh OUT ENVVAR  -> kernel traces of the bench step with ENVVAR=0 and ENVVAR=1
O=$GRAFT_REPO_ROOT/gpurun_out/$1; R=$GRAFT_REPO_ROOT; V=$2
mkdir -p $O/a $O/b
cd /tmp && export TMPDIR=/tmp
export $V=0
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/a -- python3 $R/bench.py --no-cpu-baseline --no-inference --no-networks --no-kernels > $O/a/run.log 2>&1 || exit 3
export $V=1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/b -- python3 $R/bench.py --no-cpu-baseline --no-inference --no-networks --no-kernels > $O/b/run.log 2>&1 || exit 4
