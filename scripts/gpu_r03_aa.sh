#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "glds" 2>&1 | tail -2
for v in "LBC_GLDS_4W=0" "LBC_GEMM256_CFG=5 LBC_GEMM256_MIN_TILES=1" "LBC_GEMM256_CFG=6 LBC_GEMM256_MIN_TILES=1"; do
  echo "== $v"; env $v timeout 200 python scripts/bench_ops.py 256 3 fwd,dgrad "0.c1" 2>&1 | grep -E "c1"
  env $v timeout 200 python scripts/bench_ops.py 256 3 fwd,dgrad "ds" 2>&1 | grep -E "ds"
  env $v timeout 200 python scripts/bench_ops.py 256 2 deconv "zzz" 2>&1 | grep -E "dec. +(fwd|dgrad)"
done 2>&1 | tee $R/glds_4w_ops.log
for v in "LBC_GLDS_4W=0" "LBC_GLDS_4W=1" "LBC_GLDS_4W=0" "LBC_GLDS_4W=1"; do
  echo "b256 $v: $(env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done 2>&1 | tee $R/glds_4w_ab.log
for v in "LBC_GLDS_4W=0" "LBC_GLDS_4W=1" "LBC_GLDS_4W=0" "LBC_GLDS_4W=1"; do
  echo "b32 $v: $(env $v timeout 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done 2>&1 | tee -a $R/glds_4w_ab.log
