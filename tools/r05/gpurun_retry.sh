#!/bin/bash
# usage: tools/r05/gpurun_retry.sh <timeout> '<command>'   -- retries while no GPU slot is free (exit code 3)
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
