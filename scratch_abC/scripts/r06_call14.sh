#!/bin/bash
# round 6, call 14: which of this round's kernel edits costs the 0.09 ms?  A = head with round 5's conv_hdmap.hpp and conv_c64p.hip;
# B = A + the two-zero-rows border select in conv_hdmap.hpp only
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r06 call14" > $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for rep in 1 2 3; do
  for T in scratch_prev scratch_abA scratch_abB .; do
    echo "b256 $T: $(cd $T && timeout 300 python bench.py --global-batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
cat $S
