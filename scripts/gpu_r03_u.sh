#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py tests/test_model.py -q -m gpu -x -k "c64 or hdma or fused_into_dgrad" 2>&1 | tail -2
rm -f $R/launches_u.txt
LBC_PROF_LAUNCHES=$R/launches_u.txt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
grep conv_hdma_transposed $R/launches_u.txt | awk '$3 > 7.2e10 && $3 < 7.3e10 {printf "%.1f us %.0f MB\n", $2*1e3, $4/1e6}' | tail -10
