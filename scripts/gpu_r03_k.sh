#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma" -x 2>&1 | tail -2
for L in l3.conv l4.conv l2.conv; do
  for v in "LBC_HDMAP_W4=0 LBC_HDMAP_VAR=0" "LBC_HDMAP_W4=0 LBC_HDMAP_VAR=8" "LBC_HDMAP_W4=1 LBC_HDMAP_VAR=0" "LBC_HDMAP_W4=1 LBC_HDMAP_VAR=8"; do
    echo "== $L $v: $(env $v timeout 60 python scripts/bench_ops.py 256 3 fwd $L 2>&1 | grep "$L" | tr '\n' ' ')"
  done
done 2>&1 | tee $R/asmrd_ops.log
for v in "LBC_HDMAP_W4=0 LBC_HDMAP_VAR=8" "LBC_HDMAP_W4=1 LBC_HDMAP_VAR=8"; do
  echo "== profile $v"; env $v timeout 300 python scripts/hdmap_prof.py 256 l3.conv 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $R/asmrd_prof.log
