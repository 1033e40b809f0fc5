#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 900 python -m pytest tests/test_data.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -2
timeout 900 python -m pytest tests/test_model.py -q -m gpu -x -k "training_scripts" 2>&1 | tail -2
