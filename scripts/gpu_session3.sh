#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > gpurun_out/s3_ops_new.log 2>&1; echo "ops new exit $?"
LBC_NO_GEMM256=1 timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > gpurun_out/s3_ops_old.log 2>&1; echo "ops old exit $?"
paste gpurun_out/s3_ops_new.log gpurun_out/s3_ops_old.log | head -30
timeout 300 python scripts/bench_ops.py 64 3 fwd,dgrad > gpurun_out/s3_ops_new64.log 2>&1
LBC_NO_GEMM256=1 timeout 300 python scripts/bench_ops.py 64 3 fwd,dgrad > gpurun_out/s3_ops_old64.log 2>&1
paste gpurun_out/s3_ops_new64.log gpurun_out/s3_ops_old64.log | head -30
timeout 900 python -m pytest tests/test_model.py -m gpu -q -k "declared_accuracy or baseline_batches" -rP > gpurun_out/s3_model.log 2>&1; echo "model exit $?"; tail -4 gpurun_out/s3_model.log; grep -h "bf16 vs f32\|phase-1 loss\|cosines" gpurun_out/s3_model.log | sort -u | head
LBC_HEAD_NO_MFMA=1 timeout 900 python -m pytest tests/test_model.py -m gpu -q -k "declared_accuracy" -rP > gpurun_out/s3_model_nomfma.log 2>&1; grep -h "bf16 vs f32" gpurun_out/s3_model_nomfma.log | sort -u | head -4
