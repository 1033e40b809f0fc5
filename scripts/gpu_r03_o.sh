#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "wgrad" -x 2>&1 | tail -2
timeout 900 python -m pytest tests/test_model.py -q -m gpu -x -k "gradients or parity or step" 2>&1 | tail -3
for B in 256 32; do
for v in "LBC_NO_WGRAD_DEFER=1" "LBC_NO_WGRAD_DEFER=0" "LBC_NO_WGRAD_DEFER=1" "LBC_NO_WGRAD_DEFER=0"; do
  echo "b$B $v: $(env $v timeout 300 python bench.py --global-batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done; done 2>&1 | tee $R/wgrad_defer_ab.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown $R/breakdown_defer.json 2>&1 | tail -1 | cut -c1-200
