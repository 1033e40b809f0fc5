#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for d in 0 4 5; do
  LBC_GLDS_DIAG=$d timeout 300 python scripts/bench_ops.py 256 3 fwd > $R/ops_diag$d.log 2>&1; echo "== diag $d exit $?"; grep "l2.conv\|l3.conv\|l4.conv" $R/ops_diag$d.log
done
