#!/bin/bash
# usage (on the GPU box): tools/run_gpu_steps.sh OUTDIR "cmd1" "cmd2" ...
# Runs the commands in order, each under its own time limit (first word = seconds), logging to OUTDIR/stepN.log.
# A step that ends in a fault, abort, segfault, time limit or hang (exit status > 1) ends the call: nothing more is
# started on the GPU.  Exit status 0 / 1 (e.g. failing tests) lets the next step run.
O=$1; shift
mkdir -p "$O"
i=0
for c in "$@"; do
  i=$((i+1))
  lim=${c%% *}; cmd=${c#* }
  echo "== step $i (limit ${lim}s): $cmd" | tee -a "$O/steps.log"
  timeout -k 10 "$lim" bash -c "$cmd" > "$O/step$i.log" 2>&1
  rc=$?
  echo "   exit $rc" | tee -a "$O/steps.log"
  tail -n 6 "$O/step$i.log"
  if [ $rc -gt 1 ]; then echo "stopping after step $i (exit $rc)" | tee -a "$O/steps.log"; exit $rc; fi
done
exit 0
