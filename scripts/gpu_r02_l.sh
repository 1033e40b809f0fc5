#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python scripts/bench_ops.py 256 3 fwd,fwd+bn,dgrad l1.conv > $R/l_ops_halo.log 2>&1; echo "== halo"; grep "l1.conv" $R/l_ops_halo.log
LBC_GEMM256_CFG=4 LBC_GEMM256_MIN_TILES=1 timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad l1.conv > $R/l_ops_g4.log 2>&1; echo "== glds2 512x64"; grep "l1.conv" $R/l_ops_g4.log
timeout 300 python -m pytest tests/test_kernels.py -q -m "not gpu or gpu" -k "glds and case1" -x 2>&1 | tail -2
