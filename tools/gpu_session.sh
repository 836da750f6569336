#!/bin/bash
# Runs a list of GPU steps on the gpurun box, one after the other; an ordinary failure (a failed assertion, rc < 124) lets the
# next step run, a step that had to be killed (timeout 124 / 137) or died on a signal ends the session -- never start another GPU
# step after a hang.  usage: tools/gpu_session.sh <outdir> <<< "T1|name1|cmd1\nT2|name2|cmd2..."
out=$1; mkdir -p "$out"
while IFS='|' read -r T name cmd; do
  [ -z "$name" ] && continue
  echo "=== $name (limit ${T}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$T" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"; tail -4 "$out/$name.log"
  if [ $rc -ge 124 ]; then echo "=== session stopped: $name was killed (rc=$rc)"; exit $rc; fi
done
exit 0
