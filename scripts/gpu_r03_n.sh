#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for B in 256 32; do
for v in "LBC_NO_SIDE_STREAM=0" "LBC_NO_SIDE_STREAM=1" "LBC_NO_SIDE_STREAM=0" "LBC_NO_SIDE_STREAM=1"; do
  echo "b$B $v: $(env $v timeout 300 python bench.py --global-batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done; done 2>&1 | tee $R/side_stream_ab.log
