#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 900 python -m pytest tests/test_model.py -q -m gpu -x -k "gradients or parity or step" 2>&1 | tail -3
for B in 256 32; do
for v in "LBC_NO_SIDE_STREAM=1" "LBC_NO_SIDE_STREAM=0" "LBC_NO_SIDE_STREAM=1" "LBC_NO_SIDE_STREAM=0"; do
  echo "b$B $v: $(env $v timeout 300 python bench.py --global-batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done; done 2>&1 | tee $R/wgrad_defer_side_ab.log
rm -rf $R/trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$R/trace" -o lbc -- python "$OLDPWD/bench.py" --global-batch 32 --steps 4 --warmup 2 --init-steps 2 --no-cpu-baseline --no-alt) > $R/trace.log 2>&1
T=$(find $R/trace -name "*kernel_trace.csv" | head -1)
python scripts/trace_gaps.py $T 2 > $R/trace_gaps_bs32.txt 2>&1
rm -rf $R/trace
head -2 $R/trace_gaps_bs32.txt | cut -c1-300
