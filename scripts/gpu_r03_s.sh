#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for b in 0 256 384 768 1024; do echo "== LBC_WGRAD_TR_BLOCKS=$b"; LBC_WGRAD_TR_BLOCKS=$b timeout 120 python scripts/bench_wgrad_group.py 256 2>&1 | grep -v amdgpu.ids; done | tee $R/wgrad_group_slots.log
echo "== batch 32"; timeout 120 python scripts/bench_wgrad_group.py 32 2>&1 | grep -v amdgpu.ids | tee -a $R/wgrad_group_slots.log
