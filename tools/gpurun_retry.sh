#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'   - retries while gpurun reports "no slot free" (exit 3), up to 40 tries
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
