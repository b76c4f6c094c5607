#!/bin/bash
# tools/gpurun_retry.sh <log> <timeout> : runs tools/_job.sh through gpurun, retrying while the pod's GPU slots are busy (exit code 3)
LOG=$1; TO=${2:-1500}
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $TO -- 'bash tools/_job.sh' > $LOG 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 100
done
exit 3
