#!/bin/bash
# extra evidence: (1) rocprofv3 --kernel-trace --stats of the DEFAULT bench command (teacher forward on its side stream: per-kernel durations
# include co-running kernels; the serialized twin is r03_final_rocprofv3_kernel_stats_serial_bf16.csv), (2) LDS counters of the convolution kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
rm -rf $R/prof_default
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof_default" -o lbc -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --init-steps 2 --no-cpu-baseline --no-alt) > $R/prof_default.log 2>&1
echo "prof default exit $?"; tail -1 $R/prof_default.log | cut -c1-200
find $R/prof_default -name "*kernel_trace*" -delete
rm -rf $R/pmc_lds
(cd /tmp && LBC_NO_SIDE_STREAM=1 timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$OLDPWD/$R/pmc_lds" -o lbc -- python "$OLDPWD/bench.py" --serial --steps 1 --warmup 1 --init-steps 1 --no-cpu-baseline --no-alt) > $R/pmc_lds.log 2>&1
echo "pmc lds exit $?"
find $R/pmc_lds -name "*kernel_trace*" -delete
python - <<'PY'
import csv, collections, glob, re
f = glob.glob('gpurun_out/pmc_lds/**/*counter_collection.csv', recursive=True)
if f:
    per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        per[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_LDS_IDX_ACTIVE': cnt[k] += 1
    with open('gpurun_out/pmc_lds_summary.txt', 'w') as o:
        o.write('%-72s %5s %14s %14s %9s\n' % ('kernel', 'n', 'LDS_IDX_ACTIVE', 'BANK_CONFLICT', 'conflict%'))
        for k, v in sorted(per.items(), key=lambda kv: -kv[1].get('SQ_LDS_IDX_ACTIVE', 0))[:24]:
            a, b = v.get('SQ_LDS_IDX_ACTIVE', 0), v.get('SQ_LDS_BANK_CONFLICT', 0)
            o.write('%-72s %5d %14.3e %14.3e %8.1f%%\n' % (k, cnt[k], a, b, 100 * b / a if a else 0))
    print(open('gpurun_out/pmc_lds_summary.txt').read())
PY
