#!/bin/bash
# Builder convenience: run a command on the MI355X box through gpurun; when no slot is free (exit 3: nothing ran,
# nothing charged) wait and ask again.  Any other exit code is returned as is (a failed GPU step is never re-run).
#   tools/gpu.sh [--timeout S] -- 'command'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
