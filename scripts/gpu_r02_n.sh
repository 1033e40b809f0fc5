#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "glds" -x > $R/n_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $R/n_pytest.log
for kt in 64 32; do LBC_GLDS_KT=$kt timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > $R/n_ops_$kt.log 2>&1; echo "== KT $kt"; grep "fwd\|l..conv *dgrad" $R/n_ops_$kt.log; done
