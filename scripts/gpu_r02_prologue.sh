#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 200 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma" -x > $R/pro_pytest.log 2>&1; echo "pytest exit $?"; tail -2 $R/pro_pytest.log
LBC_HDMA_PROLOGUE=1 timeout 300 python bench.py --no-cpu-baseline --no-alt --breakdown $R/pro_breakdown.json > $R/pro_bench.log 2>&1; echo "prologue bench: $(tail -1 $R/pro_bench.log | cut -c100-200)"
python - <<'PY'
import json
d=json.load(open('gpurun_out/pro_breakdown.json'))['classes']
for k in ('conv_hdma_gather','bn_apply','conv_wgrad_tr'): print(k, d.get(k))
PY
