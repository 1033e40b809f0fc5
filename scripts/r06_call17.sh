#!/bin/bash
# round 6, call 17: head (shared zero slot back, stem two-ahead only from 24 tiles per workgroup, Adam two groups per trip) against round 5's tree; full GPU test suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r06 call17" > $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 32; do
for rep in 1 2 3; do
  for T in scratch_prev .; do
    echo "b$B $T: $(cd $T && timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
done
for T in scratch_prev .; do
  (cd $T && timeout 200 python bench.py --steps 5 --warmup 3 --init-steps 2 --no-cpu-baseline --no-alt --breakdown /tmp/bd_$T$$.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/bd_$T$$.json'))['classes']; print('$T', {k: (v['launches'], round(v['ms'],3)) for k,v in d.items() if 'stem' in k or 'adam' in k or 'hdma' in k or 'c64' in k})") >> $S 2>&1
done
timeout 1800 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $S; tail -4 $R/pytest_gpu.log >> $S
cat $S
