#!/bin/bash
# round 6, call 5: floor of the MFMA stream (no reads, no requests), and the priority of the multiplying waves
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
P=$R/hdmaw_prof2.txt; echo "== $(date) conv_hdmaw_k: MFMA floor, priority" > $P
for V in "" _NOREAD_NODMA _NOPRIO _NOREAD_NOPRIO; do
  echo "--- build: hdmaw_prof$V" >> $P
  timeout 60 scripts/probe/hdmaw_prof$V >> $P 2>&1; echo "exit $?" >> $P
done
grep -E "build|launch|multiplying|loading|clock" $P | cut -c1-330
