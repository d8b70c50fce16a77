#!/bin/bash
# gpurun with a wait for a free slot: exit code 3 means "no box or slot free, nothing charged" -- try again after a
# minute (only that code; a failed or killed GPU command is never re-run).   tools/gpu.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
