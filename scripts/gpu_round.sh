#!/bin/bash
# One GPU-box session: parity tests, bench with per-kernel breakdown, rocprofv3 kernel stats.
# Everything worth keeping is written under gpurun_out/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
PHASES="${1:-test bench prof}"
echo "== $(date) phases: $PHASES" > gpurun_out/summary.txt
rocm-smi --showproductname 2>/dev/null | head -8 >> gpurun_out/summary.txt
if [[ "$PHASES" == *test* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/summary.txt
  tail -15 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
fi
if [[ "$PHASES" == *smoke* ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit $?" >> gpurun_out/summary.txt; tail -3 gpurun_out/smoke.log >> gpurun_out/summary.txt
fi
if [[ "$PHASES" == *bench* ]]; then
  timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 --breakdown gpurun_out/breakdown.json ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1
  echo "bench exit $?" >> gpurun_out/summary.txt
  tail -2 gpurun_out/bench.log >> gpurun_out/summary.txt
fi
if [[ "$PHASES" == *prof* ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o lbc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --init-steps 2 --no-cpu-baseline) > gpurun_out/prof.log 2>&1
  echo "prof exit $?" >> gpurun_out/summary.txt
  find gpurun_out/prof -name "*kernel_stats*" | head -3 >> gpurun_out/summary.txt
  # keep the (large) raw trace out of the merge budget
  find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
fi
cat gpurun_out/summary.txt
