#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "wgrad" -x 2>&1 | tail -2
timeout 200 python bench.py --no-cpu-baseline --no-alt --breakdown $R/i_breakdown.json 2>&1 | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("step", d["ms_per_step"], d["roofline"]["frac"], {k:(v["launches"], v["ms"],v["tflops"]) for k,v in d["roofline"]["by_kernel"].items()})'
python - <<'PY'
import json
d=json.load(open("gpurun_out/i_breakdown.json"))["classes"]
tot=sum(v["ms"] for v in d.values())
print("sum of kernel ms", round(tot,3))
for k,v in sorted(d.items(), key=lambda kv:-kv[1]["ms"])[:32]:
    print("  %-26s n=%3d %7.3f ms" % (k, v["launches"], v["ms"]))
PY
timeout 200 python bench.py --global-batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown $R/i_breakdown_b32.json 2>&1 | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("b32 step", d["ms_per_step"])'
python - <<'PY'
import json
d=json.load(open("gpurun_out/i_breakdown_b32.json"))["classes"]
tot=sum(v["ms"] for v in d.values()); n=sum(v["launches"] for v in d.values())
print("b32: sum of kernel ms", round(tot,3), "launches", n)
for k,v in sorted(d.items(), key=lambda kv:-kv[1]["ms"])[:24]:
    print("  %-26s n=%3d %7.3f ms" % (k, v["launches"], v["ms"]))
PY
