#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_model.py -q -m gpu -x -k "c64 or hdma or fused_into_dgrad or glds or below_the_planned or gradients" 2>&1 | tail -2
timeout 120 python scripts/bench_ops.py 256 3 fwd,dgrad l1.conv 2>&1 | grep l1.conv
rm -f $R/launches_v.txt
LBC_PROF_LAUNCHES=$R/launches_v.txt timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
grep conv_hdma_transposed $R/launches_v.txt | awk '$3 > 7.2e10 && $3 < 7.3e10 {printf "%.1f us %.0f MB\n", $2*1e3, $4/1e6}' | tail -10
