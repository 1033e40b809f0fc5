#!/bin/bash
# round 6, call 2: the wave-specialised persistent halo-staged convolution (conv_hdmaw_k: four multiplying + four loading waves) against the
# eight-wave all-purpose kernel (LBC_HDMAW=0) -- GPU parity, per-launch and step A/B on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r06 call2" > $S
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "hdma" > $R/pytest_gpu_hdmaw.log 2>&1; echo "pytest kernels exit $?" >> $S; tail -3 $R/pytest_gpu_hdmaw.log >> $S
for L in l2.conv l3.conv l4.conv; do
  for OP in fwd dgrad; do
    echo "$L $OP at 256 images, wave-specialised: $(timeout 100 python scripts/bench_ops.py 256 3 $OP $L 2>/dev/null | grep $OP | head -1) | eight-wave: $(LBC_HDMAW=0 timeout 100 python scripts/bench_ops.py 256 3 $OP $L 2>/dev/null | grep $OP | head -1)" >> $S
  done
done
timeout 900 python -m pytest tests/test_model.py tests/test_step.py -m gpu -q -x -k "bn_backward_reduce_fused or frozen or k_steps or engine_full_size or parity_at_bench" > $R/pytest_gpu_model.log 2>&1; echo "pytest model exit $?" >> $S; tail -3 $R/pytest_gpu_model.log >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 128; do
  for rep in 1 2; do
    echo "b$B eight-wave kernel (LBC_HDMAW=0): $(LBC_HDMAW=0 timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
    echo "b$B wave-specialised: $(timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
cat $S
