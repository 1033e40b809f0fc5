#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for mt in 192 48 16; do LBC_GEMM256_MIN_TILES=$mt timeout 300 python scripts/bench_ops.py 32 3 fwd,dgrad > $R/w_ops_$mt.log 2>&1; echo "== batch 32, min tiles $mt"; grep "fwd\|dgrad" $R/w_ops_$mt.log | grep -v "l1.conv\|ds "; done
for mt in 192 48 16; do LBC_GEMM256_MIN_TILES=$mt timeout 300 python bench.py --global-batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $R/w_b32_$mt.log 2>&1; echo "b32 min tiles $mt: $(tail -1 $R/w_b32_$mt.log | cut -c100-200)"; done
