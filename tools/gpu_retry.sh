#!/bin/bash
# Runs one gpurun call; when the pool has no free slot or box (exit code 3: nothing ran, nothing was charged) waits and
# asks again.  Any other exit code ends the script with it.
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 75
done
exit 3
