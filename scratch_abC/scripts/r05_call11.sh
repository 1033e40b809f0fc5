#!/bin/bash
# round 5, call 11: head forward with its A fragments requested one group ahead -- GPU parity, same-box A/B against the previous commit
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r05 call11" > $S
timeout 600 python -m pytest tests -m gpu -q -x -k "head or k_steps_bf16 or batch1" > $R/pytest_gpu_head.log 2>&1; echo "pytest exit $?" >> $S; tail -3 $R/pytest_gpu_head.log >> $S
grep -a "k-step parity bf16" $R/grad_diag.txt | tail -1 >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s', 'head_fwd', [round(v['ms'],4) for k,v in []])" 2>&1 | tail -1; }
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 32; do
  for rep in 1 2; do
    echo "b$B previous commit: $(cd scratch_prev && timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
    echo "b$B head forward one group ahead: $(timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
for T in scratch_prev .; do
  (cd $T && timeout 200 python bench.py --steps 5 --warmup 3 --init-steps 2 --no-cpu-baseline --no-alt --breakdown /tmp/bd_$$.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/bd_$$.json'))['classes']; print('$T head_fwd', d['head_fwd'], 'head_bwd_reduce', d['head_bwd_reduce']['ms'])") >> $S
done
cat $S
