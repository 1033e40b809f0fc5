#!/bin/bash
# final evidence without the PMC passes (those of run 5 stay): driver-default bench + breakdown, rocprofv3 kernel stats, full GPU test suite, smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date)" > $S
timeout 600 python bench.py --breakdown $R/breakdown_bs256_bf16.json > $R/bench_bf16.log 2>&1; echo "bench exit $?" >> $S; tail -1 $R/bench_bf16.log | cut -c1-400 >> $S
rm -rf $R/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof" -o lbc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --init-steps 2 --no-cpu-baseline --no-alt) > $R/prof.log 2>&1
echo "prof exit $?" >> $S
find $R/prof -name "*kernel_trace*" -size +20M -delete
rm -f $R/grad_diag.txt
timeout ${PYTEST_TIMEOUT:-1200} python -m pytest tests -m gpu -q --durations=5 > $R/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $S; tail -10 $R/pytest_gpu.log >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke exit $?: $(tail -1 $R/smoke.log)" >> $S
cat $S
