#!/bin/bash
# round 6, call 15: the shared zero slot (round 5's border select, on the last row of the two zero rows) against the lane's-own-bank zero, everything else equal
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r06 call15" > $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 32; do
for rep in 1 2 3; do
  for T in scratch_prev scratch_abC .; do
    echo "b$B $T: $(cd $T && timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
done
cat $S
