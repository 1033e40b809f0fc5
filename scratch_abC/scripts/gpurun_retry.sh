#!/bin/bash
# retry wrapper around gpurun for "no slot free right now" (exit 3)
T=${GPU_TIMEOUT:-1500}
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  if grep -q '"status": "transient"' gpurun_out/.last_call.json 2>/dev/null; then sleep 45; else exit $rc; fi
done
