#!/bin/bash
# gpurun with retries while no box / slot is free (exit code 3: nothing charged): tools/gpu_retry.sh TIMEOUT_S 'command'
T=$1; shift
for k in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
