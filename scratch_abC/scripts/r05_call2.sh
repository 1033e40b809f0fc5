#!/bin/bash
# round 5, call 2: information for the small-batch work -- grid-barrier probe, free-running per-kernel durations at 32 / 64 images
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r05 call2: grid-barrier probe + kernel stats at 32 / 64 images" > $S
timeout 120 scripts/probe/gridbar_probe > $R/gridbar_probe.txt 2>&1; echo "probe exit $?" >> $S; cat $R/gridbar_probe.txt >> $S
for B in 32 64; do
  rm -rf $R/prof_b$B
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof_b$B" -o lbc -- python "$OLDPWD/bench.py" --global-batch $B --steps 20 --warmup 5 --init-steps 2 --no-cpu-baseline --no-alt) > $R/prof_b$B.log 2>&1
  echo "prof b$B exit $?" >> $S
  find $R/prof_b$B -name "*kernel_trace*" -size +20M -delete
  cp $(find $R/prof_b$B -name "*kernel_stats.csv" | head -1) $R/kernel_stats_b$B.csv 2>/dev/null
done
cat $S
