#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
rm -rf $R/trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$R/trace" -o lbc -- python "$OLDPWD/bench.py" --steps 4 --warmup 2 --init-steps 2 --no-cpu-baseline --no-alt) > $R/trace.log 2>&1
T=$(find $R/trace -name "*kernel_trace.csv" | head -1)
python scripts/trace_gaps.py $T 2 > $R/trace_gaps_bs256.txt 2>&1
rm -rf $R/trace
(cd /tmp && LBC_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$R/trace" -o lbc -- python "$OLDPWD/bench.py" --steps 4 --warmup 2 --init-steps 2 --no-cpu-baseline --no-alt) > $R/trace.log 2>&1
T=$(find $R/trace -name "*kernel_trace.csv" | head -1)
python scripts/trace_gaps.py $T 2 > $R/trace_gaps_bs256_noside.txt 2>&1
rm -rf $R/trace
head -3 $R/trace_gaps_bs256.txt | cut -c1-300; head -3 $R/trace_gaps_bs256_noside.txt | cut -c1-300
