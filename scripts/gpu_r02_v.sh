#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma" -x > $R/v_pytest.log 2>&1; echo "pytest exit $?"; tail -2 $R/v_pytest.log
timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad l1.conv > $R/v_ops.log 2>&1; echo "== hdma64"; grep "l1.conv" $R/v_ops.log
LBC_NO_HDMA64=1 timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad l1.conv > $R/v_ops_old.log 2>&1; echo "== conv_halo"; grep "l1.conv" $R/v_ops_old.log
