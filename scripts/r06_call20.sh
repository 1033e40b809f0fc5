#!/bin/bash
# round 6, call 18: prep_input_u8 row-wise (coalesced 4-byte loads through LDS, dword stores), against round 5's tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r06 call20" > $S
timeout 600 python -m pytest tests -m gpu -q -x -k "stem or uint8 or engine_full_size or u8 or loader or inference or session" > $R/pytest_gpu_stem.log 2>&1; echo "pytest exit $?" >> $S; tail -2 $R/pytest_gpu_stem.log >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 32; do
for rep in 1 2 3; do
  for T in scratch_prev .; do
    echo "b$B $T: $(cd $T && timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
done
for T in scratch_prev .; do
  (cd $T && timeout 200 python bench.py --steps 5 --warmup 3 --init-steps 2 --no-cpu-baseline --no-alt --breakdown /tmp/bd_$$.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/bd_$$.json'))['classes']; print('$T', {k: (v['launches'], round(v['ms'],3)) for k,v in d.items() if 'stem' in k or 'pool' in k or 'prep' in k})") >> $S 2>&1
done
cat $S
