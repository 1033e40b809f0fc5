#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 300 python scripts/bench_ops.py 256 3 fwd,dgrad > $R/s6_ops.log 2>&1; echo "ops exit $?"; grep "l2.conv\|l3.conv\|l4.conv" $R/s6_ops.log
timeout 600 python bench.py --no-cpu-baseline --no-alt --breakdown $R/s6_breakdown_bf16.json > $R/s6_bench.log 2>&1; echo "bench exit $?"; tail -1 $R/s6_bench.log | cut -c1-260
python - <<'PY'
import json
d=json.load(open('gpurun_out/s6_breakdown_bf16.json'))['classes']
for k in ('stem_wgrad','conv_glds_gather','conv_glds_transposed'): print(k, d.get(k))
PY
rm -f $R/grad_diag.txt
timeout 1800 python -m pytest tests -m gpu -q > $R/s6_pytest.log 2>&1; echo "pytest exit $?"; tail -15 $R/s6_pytest.log
cat $R/grad_diag.txt | cut -c1-400
