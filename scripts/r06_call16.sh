#!/bin/bash
# round 6, call 16: per-kernel time of the 32-image and 256-image steps, round 5's tree against the head (rocprofv3 --stats, free-running)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for B in 32 256; do
for T in prev head; do
  D=.; [ $T = prev ] && D=scratch_prev
  rm -rf $R/prof_$T
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof_$T" -o lbc -- python "$OLDPWD/$D/bench.py" --global-batch $B --steps 20 --warmup 5 --init-steps 2 --no-cpu-baseline --no-alt) > $R/prof_$T.log 2>&1
  cp $(find $R/prof_$T -name "*kernel_stats.csv" | head -1) $R/kernel_stats_b${B}_$T.csv; rm -rf $R/prof_$T
done
done
ls $R/kernel_stats_b*
