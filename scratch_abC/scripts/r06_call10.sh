#!/bin/bash
# round 6, call 10: conv_hdmap_k (eight-wave shapes) with the fragment reads inside the MFMA gaps, against round 5's tree on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r06 call10" > $S
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "hdma" > $R/pytest_gpu_hdmap.log 2>&1; echo "pytest kernels exit $?" >> $S; tail -3 $R/pytest_gpu_hdmap.log >> $S
for L in l2.conv l3.conv l4.conv; do
  for OP in fwd dgrad; do
    echo "$L $OP at 256 images, reads in the MFMA gaps: $(timeout 100 python scripts/bench_ops.py 256 3 $OP $L 2>/dev/null | grep $OP | head -1) | round 5: $(cd scratch_prev && timeout 100 python scripts/bench_ops.py 256 3 $OP $L 2>/dev/null | grep $OP | head -1)" >> $S
  done
done
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 128; do
  for rep in 1 2; do
    echo "b$B round 5: $(cd scratch_prev && timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
    echo "b$B reads in the MFMA gaps: $(timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
cat $S
