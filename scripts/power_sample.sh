#!/bin/bash
# Samples rocm-smi (power, shader clock, temperature, power cap) every 0.5 s while the default bench runs: the evidence behind DESIGN.md section 3's
# "these launches run at the chip's power cap".  usage (on the GPU box): bash scripts/power_sample.sh > gpurun_out/power_samples.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== static"; rocm-smi --showmaxpower --showperflevel --showclocks 2>&1 | grep -v "^=\|^$" | head -30
(for i in $(seq 1 60); do echo "-- t=$i"; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -iE "power|sclk|mclk|junction|edge" | head -8; sleep 0.5; done) &
SAMPLER=$!
timeout 120 python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | cut -c1-200
kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
