#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 900 python -m pytest tests/test_model.py -q -m gpu -x -k "last_workgroup or gradients_with_frozen or fused_into_dgrad" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "hdma or c64" 2>&1 | tail -2
for B in 256 32; do
for v in "LBC_NO_FIN_FUSE=1" "LBC_NO_FIN_FUSE=0" "LBC_NO_FIN_FUSE=1" "LBC_NO_FIN_FUSE=0"; do
  echo "b$B $v: $(env $v timeout 300 python bench.py --global-batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done; done 2>&1 | tee $R/fin_fuse_ab.log
timeout 120 python scripts/bench_ops.py 256 3 fwd,dgrad l1.conv 2>&1 | grep l1.conv
