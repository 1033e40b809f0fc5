#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k glds -x > $R/d_pytest.log 2>&1; echo "pytest exit $?"; tail -4 $R/d_pytest.log
timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > $R/d_ops_v2.log 2>&1; echo "== v2 exit $?"; grep "l2.conv\|l3.conv\|l4.conv" $R/d_ops_v2.log
LBC_GLDS_V1=1 timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > $R/d_ops_v1.log 2>&1; echo "== v1 exit $?"; grep "l2.conv\|l3.conv\|l4.conv" $R/d_ops_v1.log
for c in 0 1 2 3; do LBC_GEMM256_CFG=$c timeout 300 python scripts/bench_ops.py 256 3 fwd > $R/d_ops_cfg$c.log 2>&1; echo "== v2 cfg $c"; grep "l2.conv\|l3.conv\|l4.conv" $R/d_ops_cfg$c.log; done
