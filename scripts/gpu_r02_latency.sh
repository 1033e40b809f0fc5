#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for p in bf16; do timeout 300 python scripts/diag_latency.py $p 2>&1 | grep -v amdgpu.ids | tee -a $R/s_latency.log; done
