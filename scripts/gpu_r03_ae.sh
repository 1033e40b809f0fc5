#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -x -k "hdma or c64" 2>&1 | tail -2
for v in "LBC_NO_C64P_PRE=1" "LBC_NO_C64P_PRE=0"; do
  echo "== $v: $(env $v timeout 120 python scripts/bench_ops.py 256 3 fwd+bn l1.conv 2>&1 | grep l1.conv | tr '\n' ' ')"
done
for v in "LBC_NO_C64P_PRE=1" "LBC_NO_C64P_PRE=0" "LBC_NO_C64P_PRE=1" "LBC_NO_C64P_PRE=0"; do
  echo "b256 $v: $(env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done 2>&1 | tee $R/c64p_pre_ab.log
timeout 900 python -m pytest tests/test_model.py -q -m gpu -x -k "gradients or parity or step or full_size or bf16" 2>&1 | tail -2
