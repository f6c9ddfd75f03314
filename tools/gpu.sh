#!/bin/bash
# tools/gpu.sh LOG TIMEOUT 'command' — one gpurun call; re-queues ONLY when no box / slot was free (exit 3: nothing ran, nothing charged)
if [ $# -lt 3 ] || [ "${1#-}" != "$1" ]; then
  echo "usage: tools/gpu.sh LOG TIMEOUT 'command'   (LOG must not start with '-')" >&2
  exit 64
fi
log="$1"; to="$2"; shift 2
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > "$log" 2>&1
  rc=$?
  [ $rc -ne 3 ] && break
  sleep 90
done
echo "gpurun rc=$rc" >> "$log"
exit $rc
