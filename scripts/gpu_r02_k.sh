#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "tap_fused" -x > $R/k_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $R/k_pytest.log
timeout 300 python scripts/bench_ops.py 256 3 wgrad,wgrad+bn > $R/k_ops.log 2>&1; echo "== exit $?"; grep "conv " $R/k_ops.log
