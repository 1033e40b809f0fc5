#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for v in "LBC_NO_WGRAD_TR2=1" "LBC_WGRAD_TR2_BLOCKS=128" "LBC_WGRAD_TR2_BLOCKS=256" "LBC_WGRAD_TR2_BLOCKS=512" "LBC_WGRAD_TR2_BLOCKS=1024"; do
  echo "== $v"; env $v timeout 120 python scripts/bench_ops.py 256 2 deconv,wgrad .0.c1 2>&1 | grep "wgrad"
done 2>&1 | tee $R/wgrad_tr2_sweep.log
for B in 128 64; do
for v in "LBC_NO_WGRAD_TR2=1" "LBC_WGRAD_TR2_MIN_WGS=1" "LBC_NO_WGRAD_TR2=1" "LBC_WGRAD_TR2_MIN_WGS=1"; do
  echo "b$B $v: $(env $v timeout 300 python bench.py --global-batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done; done 2>&1 | tee $R/wgrad_tr2_ab_small.log
