#!/bin/bash
# rocm-smi every 0.5 s while ONE convolution launch (layer 3 forward at batch 256, conv_hdmap_k) repeats back to back for ~12 s, then the same with zero operands:
# package power against the 1400 W cap and the shader clock under a pure MFMA load
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for F in "" zero; do
  echo "== operands '$F'"
  (for i in $(seq 1 30); do rocm-smi --showpower --showclocks 2>&1 | grep -iE "Package Power|sclk" | sed 's/GPU\[0\]\t\t: //' | paste - -; sleep 0.5; done) &
  SAMPLER=$!
  BENCH_OPS_FILL=$F BENCH_OPS_ITERS=150000 timeout 60 python scripts/bench_ops.py 256 3 fwd l3.conv 2>/dev/null | grep fwd
  kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
done
