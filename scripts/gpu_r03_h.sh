#!/bin/bash
# Round 3, GPU call 9: the persistent 64-channel kernel (conv_c64p.hip) vs conv_halo.hip -- parity, per-layer and per-step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma or c64" -x 2>&1 | tail -2
for v in "LBC_NO_HDMA64=1" "LBC_NO_HDMA64=0"; do
  echo "== l1.conv $v: $(env $v timeout 60 python scripts/bench_ops.py 256 3 fwd,dgrad l1.conv 2>&1 | grep "l1.conv" | tr '\n' ' ')"
done 2>&1 | tee $R/c64p_ops.log
for v in 1 0; do
  LBC_NO_HDMA64=$v timeout 200 python bench.py --no-cpu-baseline --no-alt --breakdown $R/c64p_breakdown_$v.json > $R/c64p_bench_$v.log 2>&1
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
line = [l for l in open("gpurun_out/c64p_bench_%s.log" % v) if l.startswith("{")][-1]
b = json.loads(line)
print("no_hdma64=%s ms_per_step %.3f roofline %.4f" % (v, b["ms_per_step"], b["roofline"]["frac"]), {k: (x["launches"], x["ms"], x["tflops"]) for k, x in b["roofline"]["by_kernel"].items() if "hdma" in k or "halo" in k})
PY
done 2>&1 | tee $R/c64p_step.log
timeout 900 python -m pytest tests/test_model.py -q -m gpu -k "fused_into_dgrad or bf16_gradients_match or test_engine_bf16_mfma_mode" 2>&1 | tail -30 > $R/c64p_pytest.log; tail -4 $R/c64p_pytest.log; cat $R/grad_diag.txt | tail -3
