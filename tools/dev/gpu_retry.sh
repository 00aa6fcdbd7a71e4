#!/bin/bash
# tools/dev/gpu_retry.sh <log name> <timeout>: run tools/dev/run1.sh on a GPU box, retrying while the pod's slots are busy
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout ${2:-1800} -- "bash tools/dev/run1.sh > gpurun_out/$1.log 2>&1; tail -40 gpurun_out/$1.log" > /tmp/$1.out 2>&1
  grep -q "status=ok\|status=fail\|status=timeout" /tmp/$1.out && break
  sleep 45
done
