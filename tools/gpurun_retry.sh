#!/bin/bash
# gpurun with retries on "no slot / no box free" (exit code 3: nothing charged).  usage: tools/gpurun_retry.sh <timeout_s> '<command>'
T=$1; shift
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
