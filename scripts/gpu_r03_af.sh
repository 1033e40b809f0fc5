#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model.py tests/test_kernels.py -q -m gpu -x -k "staged_backward_equals or stride2_tap_fused_random" 2>&1 | tail -2
