#!/bin/bash
# round 5, call 14: the stride-2 transposed launches on the persistent halo-staged kernel (conv_hdmap_k<.., MODE 2>: 2 x 2 halo, four
# accumulator sets): GPU parity, per-launch and same-box step A/B against the previous commit (per-tap kernel conv_glds2_k<.., PH>)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r05 call14" > $S
timeout 900 python -m pytest tests -m gpu -q -x -k "phased_transposed or glds or deconv or stride2 or bn_backward_reduce_fused or bf16_gradients_with_frozen or k_steps_bf16 or engine_full_size" > $R/pytest_gpu_phased.log 2>&1; echo "pytest exit $?" >> $S; tail -4 $R/pytest_gpu_phased.log >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 32; do
  for rep in 1 2; do
    echo "b$B previous commit: $(cd scratch_prev && timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
    echo "b$B phased transposed launches on conv_hdmap_k MODE 2: $(timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
echo "b64 previous commit: $(cd scratch_prev && timeout 300 python bench.py --global-batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
echo "b64 phased transposed launches on conv_hdmap_k MODE 2: $(timeout 300 python bench.py --global-batch 64 --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
for L in l2.0 l3.0 l4.0; do
  echo "$L dgrad at 256 images, head: $(timeout 100 python scripts/bench_ops.py 256 3 dgrad $L 2>/dev/null | grep dgrad | head -1) | previous: $(cd scratch_prev && timeout 100 python scripts/bench_ops.py 256 3 dgrad $L 2>/dev/null | grep dgrad | head -1)" >> $S
done
for T in scratch_prev .; do
  (cd $T && timeout 200 python bench.py --steps 5 --warmup 3 --init-steps 2 --no-cpu-baseline --no-alt --breakdown /tmp/bd_$$.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/bd_$$.json'))['classes']; print('$T', {k: (v['launches'], round(v['ms'],3)) for k,v in d.items() if k.startswith('conv_') or k=='bn_apply'})") >> $S
done
cat $S
