#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_parallel.py -q -m gpu > gpurun_out/rccl1.log 2>&1; echo "exit $?"; tail -3 gpurun_out/rccl1.log
