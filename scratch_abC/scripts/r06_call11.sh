#!/bin/bash
# round 6, call 11: is the launch power-bound?  same kernels, operands zero / half zero / random
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
P=$R/power.txt; echo "== $(date) operand values vs launch time" > $P
for F in 1 2 0; do timeout 60 scripts/probe/hdmaw_prof 10 24 256 256 256 $F 2>&1 | grep -E "fill|launch|clock" >> $P; done
timeout 60 scripts/probe/hdmaw_prof_NOREAD_NODMA 10 24 256 256 256 1 2>&1 | grep -E "launch|clock" >> $P
for F in "" relu zero; do
  for L in l3.conv l2.conv; do
    echo "conv_hdmap_k $L fwd, operands '$F': $(BENCH_OPS_FILL=$F timeout 100 python scripts/bench_ops.py 256 3 fwd $L 2>/dev/null | grep fwd | head -1)" >> $P
  done
done
cat $P
