#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "wgrad" 2>&1 | tail -2
timeout 200 python scripts/bench_wgrad_group.py 256 2>&1 | grep -v amdgpu.ids | tee $R/wgrad_group_xcd.txt
timeout 200 python scripts/bench_wgrad_group.py 32 2>&1 | grep -v amdgpu.ids | tee -a $R/wgrad_group_xcd.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
