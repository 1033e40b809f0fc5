#!/bin/bash
# round 6, call 4: conv_hdmaw_k with the fragment reads of step g + 1 inside the first three MFMA gaps of step g: cycle budget, parity, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r06 call4" > $S
P=$R/hdmaw_prof.txt; echo "== $(date) conv_hdmaw_k cycle budget (reads inside the MFMA gaps)" > $P
for V in "" _NOMFMA _NOREAD _NODMA; do
  echo "--- build: hdmaw_prof$V" >> $P
  timeout 60 scripts/probe/hdmaw_prof$V >> $P 2>&1; echo "exit $?" >> $P
done
timeout 60 scripts/probe/hdmaw_prof 20 48 128 128 256 >> $P 2>&1
timeout 60 scripts/probe/hdmaw_prof 5 12 512 512 256 >> $P 2>&1
grep -E "build|launch|multiplying|loading" $P | cut -c1-330 >> $S
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -x -k "hdma" > $R/pytest_gpu_hdmaw.log 2>&1; echo "pytest kernels exit $?" >> $S; tail -3 $R/pytest_gpu_hdmaw.log >> $S
for L in l2.conv l3.conv l4.conv; do
  for OP in fwd dgrad; do
    echo "$L $OP at 256 images, wave-specialised: $(timeout 100 python scripts/bench_ops.py 256 3 $OP $L 2>/dev/null | grep $OP | head -1) | eight-wave: $(LBC_HDMAW=0 timeout 100 python scripts/bench_ops.py 256 3 $OP $L 2>/dev/null | grep $OP | head -1)" >> $S
  done
done
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256; do
  for rep in 1 2; do
    echo "b$B eight-wave kernel (LBC_HDMAW=0): $(LBC_HDMAW=0 timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
    echo "b$B wave-specialised: $(timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
  done
done
cat $S
