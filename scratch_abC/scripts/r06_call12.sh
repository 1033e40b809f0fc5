#!/bin/bash
# round 6, call 12: stem forward with the bands of the next TWO tiles in flight; ADVICE fixes; same-box A/B against round 5's tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r06 call13" > $S
echo "(tests: see call 12)" >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 32; do
  for rep in 1 2; do
    echo "b$B round 5: $(cd scratch_prev && timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
    echo "b$B head: $(timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
for T in scratch_prev .; do
  (cd $T && timeout 200 python bench.py --steps 5 --warmup 3 --init-steps 2 --no-cpu-baseline --no-alt --breakdown /tmp/bd_$$.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/bd_$$.json'))['classes']; print('$T', {k: (v['launches'], round(v['ms'],3)) for k,v in d.items() if 'stem' in k or 'pool' in k or 'prep' in k})") >> $S
done
cat $S
