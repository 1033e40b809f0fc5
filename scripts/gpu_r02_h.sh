#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_ops.py tests/test_kernels.py -q -m gpu -k "stem or glds" -x > $R/h_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $R/h_pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-alt --breakdown $R/h_breakdown.json > $R/h_bench.log 2>&1; echo "bench exit $?"; tail -1 $R/h_bench.log | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/h_breakdown.json'))['classes']
tot=sum(v['ms'] for v in d.values()); print('instrumented total', round(tot,3))
for k,v in sorted(d.items(), key=lambda kv:-kv[1]['ms'])[:22]: print("%-26s %4d %8.3f"%(k,v['launches'],v['ms']))
PY
LBC_NO_FUSE_Z1=0 timeout 600 python bench.py --no-cpu-baseline --no-alt > $R/h_bench_fused.log 2>&1; echo "fused-everywhere: $(tail -1 $R/h_bench_fused.log | cut -c1-200)"
