#!/bin/bash
# Round 3, GPU call 2: (1) what conv_hdma_k's time is made of (LBC_HDMA_DIAG bit mask: 1 no main-loop DMA, 2 no fragment reads,
# 4 no barriers, 8 no epilogue, 16 no MFMA), (2) phase-1 loss curves per precision mode with controls, (3) layer-1 A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for L in l3.conv l2.conv l4.conv; do
  for D in 0 1 2 4 8 16 3 5 6 9 7 15 24 18; do
    echo "== $L diag $D: $(LBC_HDMA_CFG=1 LBC_HDMA_DIAG=$D timeout 60 python scripts/bench_ops.py 256 3 fwd $L 2>&1 | grep "$L" | head -1)"
  done
done 2>&1 | tee $R/hdma_diag.log
timeout 400 python scripts/diag_bf16_curve.py 200 32 $R/bf16_curves.json 2>&1 | tee $R/bf16_curves.log | tail -12
for f in "" "LBC_NO_HALO=1 LBC_NO_HDMA=1 LBC_GEMM256_CFG=4 LBC_GEMM256_MIN_TILES=1" "LBC_NO_HALO=1 LBC_HDMA_CFG=3"; do
  echo "== l1 [$f]"; env $f timeout 100 python scripts/bench_ops.py 256 3 fwd,dgrad l1 2>&1 | grep l1
done 2>&1 | tee $R/l1_ab.log
