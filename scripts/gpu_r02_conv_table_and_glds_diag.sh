#!/bin/bash
# round-2 session B: per-layer convolution table + conv_glds load-path experiments (LBC_GLDS_DIAG, timing only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 400 python scripts/bench_ops.py 256 3 fwd,fwd+bn,dgrad,wgrad,wgrad+bn > $R/ops_b256.log 2>&1; echo "ops exit $?"
for d in 1 2 3; do
  LBC_GLDS_DIAG=$d timeout 300 python scripts/bench_ops.py 256 3 fwd > $R/ops_diag$d.log 2>&1; echo "diag $d exit $?"
done
grep -v amdgpu.ids $R/ops_b256.log
for d in 1 2 3; do echo "== diag $d"; grep "l2.conv\|l3.conv\|l4.conv" $R/ops_diag$d.log; done
