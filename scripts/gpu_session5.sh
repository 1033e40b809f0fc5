#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > $R/s5_ops.log 2>&1; echo "ops exit $?"; grep "l2.conv\|l3.conv\|l4.conv" $R/s5_ops.log
LBC_GEMM256_CFG=1 timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > $R/s5_ops_cfg1.log 2>&1; grep "l2.conv" $R/s5_ops_cfg1.log
timeout 600 python bench.py --no-cpu-baseline --no-alt --breakdown $R/s5_breakdown_bf16.json > $R/s5_bench.log 2>&1; echo "bench exit $?"; tail -1 $R/s5_bench.log | cut -c1-260
rm -f $R/grad_diag.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $R/s5_pytest.log 2>&1; echo "pytest exit $?"; tail -12 $R/s5_pytest.log
