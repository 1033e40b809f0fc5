#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python bench.py --global-batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown $R/breakdown_b32_bf16.json > $R/bench_b32_bf16.log 2>&1; echo "b32: $(tail -1 $R/bench_b32_bf16.log | cut -c100-200)"
timeout 300 python bench.py --global-batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $R/bench_b64_bf16.log 2>&1; echo "b64: $(tail -1 $R/bench_b64_bf16.log | cut -c100-200)"
timeout 300 python bench.py --global-batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $R/bench_b128_bf16.log 2>&1; echo "b128: $(tail -1 $R/bench_b128_bf16.log | cut -c100-200)"
rm -f $R/grad_diag.txt
timeout 900 python -m pytest tests/test_model.py tests/test_kernels.py -q -m gpu -k "bf16 or hdma or fused_into or declared" > $R/x_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $R/x_pytest.log
