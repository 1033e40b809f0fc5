#!/bin/bash
# round 6, call 9: priority of the loading waves (s_setprio 1 / 3) in conv_hdmaw_k
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
P=$R/hdmaw_prof3.txt; echo "== $(date) conv_hdmaw_k: loader priority" > $P
for V in "" _LPRIO1 _LPRIO3; do
  echo "--- build: hdmaw_prof$V" >> $P
  timeout 60 scripts/probe/hdmaw_prof$V >> $P 2>&1; echo "exit $?" >> $P
  timeout 60 scripts/probe/hdmaw_prof$V 20 48 128 128 256 >> $P 2>&1
done
grep -E "build|launch|multiplying|loading|clock" $P | cut -c1-330
