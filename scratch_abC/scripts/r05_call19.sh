#!/bin/bash
# the driver's round-end sequence on a fresh box, wall-clock timed: smoke(), then the default bench command
mkdir -p gpurun_out/call19
{ time timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" ; } > gpurun_out/call19/smoke.log 2>&1
{ time timeout 400 python bench.py ; } > gpurun_out/call19/bench_default.log 2>&1
tail -4 gpurun_out/call19/smoke.log; tail -5 gpurun_out/call19/bench_default.log | cut -c1-400
