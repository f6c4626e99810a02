#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 / "transient": nothing is charged).
# usage: tools/gpurun_retry.sh <timeout seconds> '<command>'
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1)
  rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 75; continue; fi
  echo "$out"
  exit $rc
done
echo "$out"; exit 3
