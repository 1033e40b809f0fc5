#!/bin/bash
# round 5, call 18: rocprofv3 --kernel-trace --stats of the driver's default bench command (side streams on: durations of co-running kernels
# are inflated; the serialized run of gpu_evidence.sh is the one per-kernel numbers are quoted from)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
rm -rf $R/prof_default
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof_default" -o lbc -- python "$OLDPWD/bench.py" --no-cpu-baseline --no-alt) > $R/prof_default.log 2>&1
echo "exit $?"; tail -1 $R/prof_default.log | cut -c1-200
cp $(find $R/prof_default -name "*kernel_stats.csv" | head -1) $R/kernel_stats_default_cmd.csv; find $R/prof_default -name "*kernel_trace*" -delete
head -5 $R/kernel_stats_default_cmd.csv | cut -c1-160
