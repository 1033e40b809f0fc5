#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma" -x > $R/u_pytest.log 2>&1; echo "pytest exit $?"; tail -2 $R/u_pytest.log
for c in -1 0; do LBC_HDMA_CFG=$c timeout 300 python scripts/bench_ops.py 256 3 fwd conv > $R/u_ops_cfg$c.log 2>&1; echo "== hdma cfg $c"; grep "l[234].conv" $R/u_ops_cfg$c.log; done
timeout 600 python bench.py --no-cpu-baseline --no-alt --breakdown $R/u_breakdown.json > $R/u_bench.log 2>&1; echo "bench exit $?"; tail -1 $R/u_bench.log | cut -c1-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/u_breakdown.json'))['classes']
tot=sum(v['ms'] for v in d.values()); print('instrumented total', round(tot,3))
for k,v in sorted(d.items(), key=lambda kv:-kv[1]['ms'])[:12]: print("%-26s %4d %8.3f"%(k,v['launches'],v['ms']))
PY
