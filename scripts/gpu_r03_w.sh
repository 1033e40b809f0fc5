#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "stride2_tap_fused or hdma or glds" 2>&1 | tail -3
for v in "LBC_NO_WGRAD_TR2=1" "LBC_NO_WGRAD_TR2=0"; do
  echo "== $v"; env $v timeout 120 python scripts/bench_ops.py 256 3 wgrad .0.c1 2>&1 | grep "c1"
done 2>&1 | tee $R/wgrad_tr2_ops.log
for B in 256 32; do
for v in "LBC_NO_WGRAD_TR2=1" "LBC_NO_WGRAD_TR2=0" "LBC_NO_WGRAD_TR2=1" "LBC_NO_WGRAD_TR2=0"; do
  echo "b$B $v: $(env $v timeout 300 python bench.py --global-batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done; done 2>&1 | tee $R/wgrad_tr2_ab.log
