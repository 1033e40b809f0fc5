#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_model.py -q -m gpu -k "bf16 or declared or inference" -x > $R/ph_pytest.log 2>&1; echo "pytest exit $?"; tail -2 $R/ph_pytest.log
LBC_GEMM256_CFG=4 timeout 300 python scripts/bench_ops.py 256 3 dgrad l2.0.c1 > $R/ph_ops4.log 2>&1; echo "== l2.0.c1 dgrad, 512x64 phased"; grep "c1" $R/ph_ops4.log
timeout 600 python bench.py --no-cpu-baseline --no-alt --breakdown $R/ph_breakdown.json > $R/ph_bench.log 2>&1; echo "bench exit $?"; tail -1 $R/ph_bench.log | cut -c100-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/ph_breakdown.json'))['classes']
tot=sum(v['ms'] for v in d.values()); print('instrumented total', round(tot,3))
for k,v in sorted(d.items(), key=lambda kv:-kv[1]['ms'])[:14]: print("%-26s %4d %8.3f"%(k,v['launches'],v['ms']))
PY
