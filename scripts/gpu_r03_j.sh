#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma" -x 2>&1 | tail -2
for L in l3.conv l4.conv l2.conv; do
  for v in "LBC_HDMA_CFG=-1" "LBC_NO_HDMA=1"; do
    echo "== b32 $L $v: $(env $v timeout 60 python scripts/bench_ops.py 32 3 fwd,dgrad $L 2>&1 | grep "$L" | tr '\n' ' ')"
  done
done 2>&1 | tee $R/small_ops.log
for B in 32 64; do
timeout 200 python bench.py --global-batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown $R/j_breakdown_b$B.json 2>&1 | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("step", d["config"]["global_batch"], d["ms_per_step"])'
done
python - <<'PY'
import json
d=json.load(open("gpurun_out/j_breakdown_b32.json"))["classes"]
tot=sum(v["ms"] for v in d.values()); n=sum(v["launches"] for v in d.values())
print("b32: sum of kernel ms", round(tot,3), "launches", n)
for k,v in sorted(d.items(), key=lambda kv:-kv[1]["ms"])[:16]:
    print("  %-26s n=%3d %7.3f ms" % (k, v["launches"], v["ms"]))
PY
