#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python scripts/diag_graph.py > $R/s7_graph.log 2>&1; echo "graph diag exit $?"; tail -6 $R/s7_graph.log
timeout 600 python scripts/diag_bf16_train.py 32 > $R/s7_bf16train.log 2>&1; echo "bf16 train diag exit $?"; grep -v amdgpu.ids $R/s7_bf16train.log | cut -c1-900
