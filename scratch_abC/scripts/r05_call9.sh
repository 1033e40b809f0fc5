#!/bin/bash
# round 5, call 9: XCD-major contiguous tile ranges in the stem forward and the pool passes (L2 locality): GPU parity tests, same-box A/B
# against the previous commit (scratch_prev/), and the FETCH_SIZE pass for the three kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r05 call9" > $S
timeout 900 python -m pytest tests -m gpu -q -x -k "stem or pool or engine_full_size or forward_parity or k_steps_bf16" > $R/pytest_gpu_stem.log 2>&1; echo "pytest exit $?" >> $S; tail -4 $R/pytest_gpu_stem.log >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 32; do
  for rep in 1 2; do
    echo "b$B previous commit: $(cd scratch_prev && timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
    echo "b$B XCD-major stem / pool ranges: $(timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
rm -rf $R/pmcF
(cd /tmp && LBC_NO_SIDE_STREAM=1 timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OLDPWD/$R/pmcF" -o lbc -- python "$OLDPWD/bench.py" --serial --steps 1 --warmup 1 --init-steps 1 --no-cpu-baseline --no-alt) > $R/pmcF.log 2>&1
python - <<'PY' >> $S
import csv, glob, collections
f = glob.glob('gpurun_out/pmcF/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if any(k in n for k in ('stem_fwd_rows', 'bn_relu_maxpool_fwd', 'maxpool_relu_bwd')) and r['Counter_Name'] == 'FETCH_SIZE':
        agg[n[:60]][0] += 1; agg[n[:60]][1] += float(r['Counter_Value'])
for k, (n, v) in agg.items():
    print('FETCH_SIZE (KiB) x 2 per launch: %-60s n=%d  %.1f MB' % (k, n, v * 2 * 1024 / n / 1e6))
PY
find $R/pmcF -name "*kernel_trace*" -delete
cat $S
