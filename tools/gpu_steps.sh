#!/bin/bash
# Runs the GPU steps given as arguments ("name::timeout_s::command") one after the other on the GPU box, every step under its own
# `timeout -k 10`, stdout+stderr of a step in gpurun_out/<name>.log.  A step that FAILS (a red test) does not stop the list; a step
# that TIMES OUT or is killed does: nothing further is started on a GPU that may be hung.
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%::*}; rest=${spec#*::}; tmo=${rest%%::*}; cmd=${rest#*::}
  echo "=== $name (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc $(( $(date +%s) - start ))s"
  tail -n 4 "gpurun_out/$name.log" | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out: stopping"; exit $rc; fi
done
exit 0
