#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
rm -rf $R/prof1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof1" -o b1 -- python "$OLDPWD/scripts/diag_graph.py") > $R/r_diag.log 2>&1; echo "exit $?"; grep -v amdgpu.ids $R/r_diag.log | tail -8
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof1/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]: print("%-90s %6s calls  avg %10.1f us  total %8.2f ms"%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
find $R/prof1 -name "*kernel_trace*" -delete
