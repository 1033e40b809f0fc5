#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 900 python -m pytest tests/test_model.py -q -m gpu -k "fused_into_dgrad or bf16" -x > $R/q_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $R/q_pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-alt --breakdown $R/q_breakdown.json > $R/q_bench.log 2>&1; echo "bench exit $?"; tail -1 $R/q_bench.log | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/q_breakdown.json'))['classes']
tot=sum(v['ms'] for v in d.values()); print('instrumented total', round(tot,3))
for k,v in sorted(d.items(), key=lambda kv:-kv[1]['ms'])[:14]: print("%-26s %4d %8.3f"%(k,v['launches'],v['ms']))
PY
