#!/bin/bash
# round 6, call 1: border lanes of the halo-staged convolutions read the zero at their own row's bank position (two zero rows) instead of
# one shared zero slot: LDS bank conflicts of conv_hdmap_k / conv_c64p_k before / after (PMC), per-launch and step A/B on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r06 call1" > $S
timeout 900 python -m pytest tests -m gpu -q -x -k "hdma or c64 or halo or phased" > $R/pytest_gpu_zero_rows.log 2>&1; echo "pytest exit $?" >> $S; tail -3 $R/pytest_gpu_zero_rows.log >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 32; do
  for rep in 1 2; do
    echo "b$B previous round: $(cd scratch_prev && timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
    echo "b$B zero rows on the lane's own banks: $(timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
for L in l1.conv l2.conv l3.conv l4.conv; do
  for OP in fwd dgrad; do
    echo "$L $OP at 256 images, head: $(timeout 100 python scripts/bench_ops.py 256 3 $OP $L 2>/dev/null | grep $OP | head -1) | previous: $(cd scratch_prev && timeout 100 python scripts/bench_ops.py 256 3 $OP $L 2>/dev/null | grep $OP | head -1)" >> $S
  done
done
for L in l3.conv l4.conv; do
  echo "$L fwd at 32 images, head: $(timeout 100 python scripts/bench_ops.py 32 3 fwd $L 2>/dev/null | grep fwd | head -1) | previous: $(cd scratch_prev && timeout 100 python scripts/bench_ops.py 32 3 fwd $L 2>/dev/null | grep fwd | head -1)" >> $S
done
rm -rf $R/pmc4
(cd /tmp && LBC_NO_SIDE_STREAM=1 timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OLDPWD/$R/pmc4" -o lbc -- python "$OLDPWD/bench.py" --serial --steps 1 --warmup 1 --init-steps 1 --no-cpu-baseline --no-alt) > $R/pmc4.log 2>&1
echo "pmc4 exit $?" >> $S
find $R/pmc4 -name "*kernel_trace*" -delete
PYTHONPATH=scripts python scripts/pmc_lds.py $(find $R/pmc4 -name "*counter_collection.csv" | head -1) > $R/pmc_lds_conflicts.txt 2>&1
head -24 $R/pmc_lds_conflicts.txt >> $S
cat $S
