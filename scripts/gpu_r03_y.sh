#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for B in 128 64 256; do
for v in "LBC_WGRAD_TR_LINEAR=1" "LBC_WGRAD_TR_LINEAR=0" "LBC_WGRAD_TR_LINEAR=1" "LBC_WGRAD_TR_LINEAR=0"; do
  echo "b$B $v: $(env $v timeout 300 python bench.py --global-batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done; done 2>&1 | tee $R/wgrad_xcd_ab.log
timeout 300 python bench.py --global-batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown $R/breakdown_b128_now.json 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
