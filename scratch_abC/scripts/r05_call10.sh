#!/bin/bash
# round 5, call 10: why did the bf16 k-step test's stem-weight gradient leave its bound after the stem / pool reordering?  Per-step errors of
# both trees (bounds relaxed), then the phase-fastest mapping of the phased transposed launches: tests + A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r05 call10" > $S
for T in scratch_prev .; do
  echo "--- tree $T" >> $S
  (cd $T && LBC_TEST_VERBOSE_STEP=1 timeout 300 python -c "
import torch, sys
sys.path.insert(0, '.')
from tests.test_step import _k_steps
_k_steps(torch.device('cuda', 0), 1, False, 3, 8, 10.0, 10.0, precision='bf16', fwd_tol=1.0, stat_rtol=1.0)
" 2>&1 | grep -a "k-step\|Error" | head -40) >> $S
done
timeout 600 python -m pytest tests -m gpu -q -x -k "glds or deconv or stride2" > $R/pytest_gpu_glds.log 2>&1; echo "pytest (glds) exit $?" >> $S; tail -3 $R/pytest_gpu_glds.log >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for rep in 1 2; do
  echo "b256 previous commit: $(cd scratch_prev && timeout 300 python bench.py --global-batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  echo "b256 head (stem / pool XCD-major + phase-fastest transposed launches): $(timeout 300 python bench.py --global-batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
done
timeout 200 python scripts/bench_ops.py 256 3 dgrad l2.0 > $R/per_shape_l20_head.txt 2>&1; (cd scratch_prev && timeout 200 python scripts/bench_ops.py 256 3 dgrad l2.0) > $R/per_shape_l20_prev.txt 2>&1
echo "layer2.0.conv1 input gradient at 256 images, head: $(grep dgrad $R/per_shape_l20_head.txt | head -1) | previous: $(grep dgrad $R/per_shape_l20_prev.txt | head -1)" >> $S
cat $S
