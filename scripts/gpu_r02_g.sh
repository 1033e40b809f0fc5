#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > $R/g_ops.log 2>&1; echo "== exit $?"; grep "l2.conv\|l3.conv\|l4.conv" $R/g_ops.log
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k glds -x > $R/g_pytest.log 2>&1; echo "pytest exit $?"; tail -2 $R/g_pytest.log
