#!/bin/bash
# First GPU call of round 3: is the read-ahead variant of the LDS-DMA convolutions (LBC_HDMA_EARLY=1)
# correct on hardware, and what does it buy?  ~4 GPU-minutes.
#   gpurun --timeout 420 -- 'bash scripts/gpu_r03_early.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
# 1. parity on the GPU: the bit-identity sub-checks of the kernel tests (real layer shapes) with the variant switched on
LBC_TEST_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma or glds" -x > $R/early_pytest.log 2>&1
echo "pytest exit $?"; tail -2 $R/early_pytest.log
# 2. the step, default and read-ahead, with the per-kernel-class breakdown of one instrumented step
timeout 200 python bench.py --no-cpu-baseline --no-alt --breakdown $R/early_breakdown_default.json > $R/early_bench_default.log 2>&1
LBC_HDMA_EARLY=1 timeout 200 python bench.py --no-cpu-baseline --no-alt --breakdown $R/early_breakdown_on.json > $R/early_bench_on.log 2>&1
python - <<'PY'
import json
for tag in ("default", "on"):
    try:
        line = [l for l in open("gpurun_out/early_bench_%s.log" % tag) if l.startswith("{")][-1]
        b = json.loads(line)
        d = json.load(open("gpurun_out/early_breakdown_%s.json" % tag))["classes"]
        print(tag, "ms_per_step", b["ms_per_step"], "roofline", b["roofline"]["frac"])
        for k in ("conv_hdma_gather", "conv_hdma_transposed", "conv_glds_gather", "conv_glds_transposed"):
            v = d.get(k)
            if v:
                print("   %-22s n=%3d %7.3f ms %6.0f TF/s" % (k, v["launches"], v["ms"], v["gflop"] / v["ms"]))
    except Exception as e:
        print(tag, "unreadable:", e)
PY
# 3. per-layer timings of the 3x3 stride-1 layers, both ways (scripts/bench_ops.py: batch, modes, ops, layer filter)
timeout 120 python scripts/bench_ops.py 256 2 fwd,dgrad > $R/early_ops_default.log 2>&1
LBC_HDMA_EARLY=1 timeout 120 python scripts/bench_ops.py 256 2 fwd,dgrad > $R/early_ops_on.log 2>&1
paste <(grep -E "layer|fwd|dgrad" $R/early_ops_default.log | head -40) <(grep -E "layer|fwd|dgrad" $R/early_ops_on.log | head -40) | cut -c1-220
# 4. layer 1 (64 -> 64 channels, 2.4 ms of the step in conv3x3_c64_k): the 512 x 64 per-tap LDS-DMA shape lost to the halo kernel
#    in round 2 (0.187 vs 0.126 ms); with read-ahead it may not (LBC_NO_HALO=1 hands layer 1 to conv_glds, cfg 4 pinned)
timeout 100 python scripts/bench_ops.py 256 2 fwd,dgrad layer1 > $R/early_l1_halo.log 2>&1
LBC_NO_HALO=1 LBC_NO_HDMA=1 LBC_GEMM256_CFG=4 LBC_GEMM256_MIN_TILES=1 timeout 100 python scripts/bench_ops.py 256 2 fwd,dgrad layer1 > $R/early_l1_glds.log 2>&1
LBC_NO_HALO=1 LBC_NO_HDMA=1 LBC_GEMM256_CFG=4 LBC_GEMM256_MIN_TILES=1 LBC_HDMA_EARLY=1 timeout 100 python scripts/bench_ops.py 256 2 fwd,dgrad layer1 > $R/early_l1_glds_early.log 2>&1
for f in early_l1_halo early_l1_glds early_l1_glds_early; do echo "== $f"; grep -E "layer1" $R/$f.log | head -6 | cut -c1-160; done
