#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "stride2_tap_fused" 2>&1 | tail -2
timeout 120 python scripts/bench_ops.py 256 2 deconv,wgrad .0.c1 2>&1 | grep "wgrad" | tee $R/wgrad_tr2_pitch160.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
