#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python scripts/hdmap_prof.py 256 l2.conv l3.conv l4.conv 2>&1 | grep -v amdgpu.ids | tee $R/hdmap_prof.log
rm -f $R/grad_diag.txt
timeout 900 python -m pytest tests/test_model.py -q -m gpu -k "bf16_gradients_match or bf16_mode_declared" 2>&1 | tail -60 > $R/parity_pytest2.log
cat $R/grad_diag.txt; grep -E "Error|passed|failed" $R/parity_pytest2.log | head
timeout 300 python -m pytest tests/test_data.py -q -m gpu 2>&1 | tail -3
