#!/bin/bash
# round 6, call 7: per-template kernel durations of the step with the wave-specialised kernel and with the eight-wave kernel (rocprofv3 --stats)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for V in 1 0; do
  rm -rf $R/prof$V
  (cd /tmp && LBC_HDMAW=$V LBC_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof$V" -o lbc -- python "$OLDPWD/bench.py" --serial --steps 3 --warmup 1 --init-steps 2 --no-cpu-baseline --no-alt) > $R/prof$V.log 2>&1
  cp $(find $R/prof$V -name "*kernel_stats.csv" | head -1) $R/kernel_stats_hdmaw$V.csv
  rm -rf $R/prof$V
done
head -40 $R/kernel_stats_hdmaw1.csv | cut -c1-200
