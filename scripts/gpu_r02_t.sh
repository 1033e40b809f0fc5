#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma" -x > $R/t_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $R/t_pytest.log
timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad conv > $R/t_ops_hdma.log 2>&1; echo "== hdma (default pick)"; grep "l[234].conv" $R/t_ops_hdma.log
for c in 1 2; do LBC_HDMA_CFG=$c timeout 300 python scripts/bench_ops.py 256 3 fwd conv > $R/t_ops_cfg$c.log 2>&1; echo "== hdma cfg $c"; grep "l[234].conv" $R/t_ops_cfg$c.log; done
LBC_NO_HDMA=1 timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad conv > $R/t_ops_glds.log 2>&1; echo "== glds2"; grep "l[234].conv" $R/t_ops_glds.log
