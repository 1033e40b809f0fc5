#!/bin/bash
# round-2 final evidence: driver-default bench + breakdown, batch 32, rocprofv3 kernel stats, PMC passes, full GPU test suite, smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date)" > $S
timeout 600 python bench.py --breakdown $R/breakdown_bs256_bf16.json > $R/bench_bf16.log 2>&1; echo "bench exit $?" >> $S; tail -1 $R/bench_bf16.log | cut -c1-400 >> $S
timeout 300 python bench.py --global-batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown $R/breakdown_b32_bf16.json > $R/bench_b32_bf16.log 2>&1
echo "b32: $(tail -1 $R/bench_b32_bf16.log | cut -c1-260)" >> $S
rm -rf $R/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof" -o lbc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --init-steps 2 --no-cpu-baseline --no-alt) > $R/prof.log 2>&1
echo "prof exit $?" >> $S
find $R/prof -name "*kernel_trace*" -size +20M -delete
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf $R/pmc$i
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OLDPWD/$R/pmc$i" -o lbc -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --init-steps 1 --no-cpu-baseline --no-alt) > $R/pmc$i.log 2>&1
  echo "pmc$i ($ctrs) exit $?" >> $S
  find $R/pmc$i -name "*kernel_trace*" -delete
done
python scripts/pmc_summary.py $(find $R/pmc1 -name "*counter_collection.csv" | head -1) $(find $R/pmc2 -name "*counter_collection.csv" | head -1) $(find $R/pmc3 -name "*counter_collection.csv" | head -1) > $R/pmc_summary.txt 2>&1
rm -f $R/grad_diag.txt
timeout ${PYTEST_TIMEOUT:-1200} python -m pytest tests -m gpu -q --durations=5 > $R/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $S; tail -10 $R/pytest_gpu.log >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke exit $?: $(tail -1 $R/smoke.log)" >> $S
cat $S
