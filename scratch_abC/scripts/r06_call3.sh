#!/bin/bash
# round 6, call 3: in-kernel cycle budget of conv_hdmaw_k (s_memtime sums per wave) + timing experiments on the layer-3 shape at batch 256
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/hdmaw_prof.txt; echo "== $(date) r06 call3: conv_hdmaw_k cycle budget" > $S
for V in "" _NOMFMA _NOREAD _NODMA _NOBAR; do
  echo "--- build: hdmaw_prof$V" >> $S
  timeout 60 scripts/probe/hdmaw_prof$V >> $S 2>&1; echo "exit $?" >> $S
done
echo "--- layer 2 (20 x 48, 128 channels) and layer 4 (5 x 12, 512 channels), full kernel" >> $S
timeout 60 scripts/probe/hdmaw_prof 20 48 128 128 256 >> $S 2>&1
timeout 60 scripts/probe/hdmaw_prof 5 12 512 512 256 >> $S 2>&1
cat $S
