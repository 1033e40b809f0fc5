#!/bin/bash
# round-2 session: new conv kernel parity + A/B timing, the failed end-to-end cases, per-tensor gradient diagnosis at N=64
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x -k "glds" > gpurun_out/s2_glds.log 2>&1; echo "glds tests exit $?"; tail -5 gpurun_out/s2_glds.log
timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > gpurun_out/s2_ops_new.log 2>&1; echo "ops new exit $?"
LBC_NO_GEMM256=1 timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > gpurun_out/s2_ops_old.log 2>&1; echo "ops old exit $?"
paste gpurun_out/s2_ops_new.log gpurun_out/s2_ops_old.log | head -30
timeout 600 python scripts/diag_grads.py image resnet34 160 384 64 f32 > gpurun_out/s2_diag64.log 2>&1; echo "diag exit $?"; grep -c "<<<" gpurun_out/s2_diag64.log; head -40 gpurun_out/s2_diag64.log
timeout 1200 python -m pytest tests/test_kernels.py -m gpu -q -k "deconv_fwd_dgrad_wgrad or test_conv_wgrad or test_conv_fwd or test_conv_dgrad" > gpurun_out/s2_kern64.log 2>&1; echo "kern64 exit $?"; tail -8 gpurun_out/s2_kern64.log
timeout 900 python -m pytest tests/test_model.py -m gpu -q -k "native_trainer or declared_accuracy or bf16_mfma_mode" -rP > gpurun_out/s2_model.log 2>&1; echo "model exit $?"; tail -8 gpurun_out/s2_model.log; grep -h "bf16 vs f32\|phase-1 loss" gpurun_out/s2_model.log | head
