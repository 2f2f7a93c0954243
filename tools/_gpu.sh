#!/bin/bash
# retry wrapper: gpurun exits 3 when no slot is free (nothing charged)
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
