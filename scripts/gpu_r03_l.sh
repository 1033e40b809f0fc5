#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 400 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma or c64" -x 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown $R/breakdown_mid.json 2>&1 | tail -1 | cut -c1-600
LBC_HDMAP_VAR=16 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | cut -c1-200
