#!/bin/bash
# Round 3, GPU call 5: the persistent halo-staged convolution (conv_hdmap.hpp) -- parity on real shapes, per-layer and per-step A/B
# against the one-tile-per-workgroup kernel -- and the round's new parity tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma" -x 2>&1 | tail -3
for L in l2.conv l3.conv l4.conv; do
  for v in 0 1; do
    echo "== $L no_persist=$v: $(LBC_NO_HDMA_PERSIST=$v timeout 60 python scripts/bench_ops.py 256 3 fwd,dgrad $L 2>&1 | grep "$L" | tr '\n' ' ')"
  done
done 2>&1 | tee $R/hdmap_ops.log
for v in 0 1; do
  LBC_NO_HDMA_PERSIST=$v timeout 200 python bench.py --no-cpu-baseline --no-alt --breakdown $R/hdmap_breakdown_$v.json > $R/hdmap_bench_$v.log 2>&1
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
line = [l for l in open("gpurun_out/hdmap_bench_%s.log" % v) if l.startswith("{")][-1]
b = json.loads(line)
print("no_persist=%s ms_per_step %.3f roofline %.4f" % (v, b["ms_per_step"], b["roofline"]["frac"]), {k: (x["ms"], x["tflops"]) for k, x in b["roofline"]["by_kernel"].items() if "hdma" in k})
PY
done 2>&1 | tee $R/hdmap_step.log
rm -f $R/grad_diag.txt
timeout 1200 python -m pytest tests/test_model.py -q -m gpu -k "frozen_decisions or bf16_gradients_match or bf16_mode_declared or fused_into_dgrad" 2>&1 | tail -25 > $R/parity_pytest.log
cat $R/grad_diag.txt; tail -12 $R/parity_pytest.log
