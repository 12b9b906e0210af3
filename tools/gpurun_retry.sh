#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout> '<command>'  -- retries while the pod's GPU slots are busy (exit code 3)
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2" > /tmp/gpurun_last.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then cat /tmp/gpurun_last.log; exit $rc; fi
  sleep 90
done
cat /tmp/gpurun_last.log; exit 3
