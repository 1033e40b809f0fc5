#!/bin/bash
# One GPU-box session: parity tests, bench with per-kernel breakdown, rocprofv3 kernel stats.
# Everything worth keeping is written under gpurun_out/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
PHASES="${1:-test bench prof}"
echo "== $(date) phases: $PHASES" > gpurun_out/summary.txt
rocm-smi --showproductname 2>/dev/null | head -8 >> gpurun_out/summary.txt
if [[ "$PHASES" == *test* ]]; then
  timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> gpurun_out/summary.txt
  tail -15 gpurun_out/pytest_gpu.log >> gpurun_out/summary.txt
fi
if [[ "$PHASES" == *smoke* ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit $?" >> gpurun_out/summary.txt; tail -3 gpurun_out/smoke.log >> gpurun_out/summary.txt
fi
if [[ "$PHASES" == *bench* ]]; then
  timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 --breakdown gpurun_out/breakdown.json ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1
  echo "bench exit $?" >> gpurun_out/summary.txt
  tail -2 gpurun_out/bench.log >> gpurun_out/summary.txt
fi
if [[ "$PHASES" == *small* ]]; then
  # per-GPU load of the 8-GPU run (32 images / GPU) on one GPU: predicts strong-scaling efficiency
  for dt in bf16 bf16_mfma f32; do
    timeout 600 python bench.py --dtype $dt --global-batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown gpurun_out/breakdown_b32_$dt.json > gpurun_out/bench_b32_$dt.log 2>&1
    echo "bench b32 $dt: $(tail -1 gpurun_out/bench_b32_$dt.log | cut -c1-260)" >> gpurun_out/summary.txt
  done
fi
if [[ "$PHASES" == *mfma* ]]; then
  timeout 900 python bench.py --dtype bf16_mfma --steps ${BENCH_STEPS:-10} --warmup 3 --no-cpu-baseline --breakdown gpurun_out/breakdown_bf16_mfma.json > gpurun_out/bench_bf16_mfma.log 2>&1
  echo "bench bf16_mfma exit $?" >> gpurun_out/summary.txt
  tail -1 gpurun_out/bench_bf16_mfma.log | cut -c1-900 >> gpurun_out/summary.txt
fi
if [[ "$PHASES" == *act* ]]; then
  # default mixed precision: bf16 MFMA operands + bf16 activation storage
  timeout 900 python bench.py --dtype bf16 --steps ${BENCH_STEPS:-10} --warmup 3 --no-cpu-baseline --no-alt --breakdown gpurun_out/breakdown_bf16.json > gpurun_out/bench_bf16.log 2>&1
  echo "bench bf16 exit $?" >> gpurun_out/summary.txt
  tail -1 gpurun_out/bench_bf16.log | cut -c1-900 >> gpurun_out/summary.txt
  timeout 600 python bench.py --dtype bf16 --global-batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown gpurun_out/breakdown_b32_bf16.json > gpurun_out/bench_b32_bf16.log 2>&1
  echo "bench b32 bf16: $(tail -1 gpurun_out/bench_b32_bf16.log | cut -c1-260)" >> gpurun_out/summary.txt
fi
if [[ "$PHASES" == *diag* ]]; then
  for v in NONE LBC_NO_FUSE_Z1 LBC_NO_DGRAD_WT; do
    env $v=1 timeout 600 python scripts/diag_grads.py ${DIAG_ARGS:-birdview resnet18 192 192 4} > gpurun_out/diag_$v.log 2>&1
    echo "diag $v exit $?" >> gpurun_out/summary.txt
  done
fi
if [[ "$PHASES" == *ab* ]]; then
  for v in LBC_NO_FUSE_Z1 LBC_NO_DGRAD_WT; do
    env $v=1 timeout 600 python bench.py --steps 5 --warmup 2 --init-steps 10 --no-cpu-baseline --breakdown gpurun_out/breakdown_$v.json > gpurun_out/bench_$v.log 2>&1
    echo "ab $v: $(tail -1 gpurun_out/bench_$v.log | cut -c1-160)" >> gpurun_out/summary.txt
  done
fi
if [[ "$PHASES" == *prof* ]]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o lbc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --init-steps 2 --no-cpu-baseline --no-alt) > gpurun_out/prof.log 2>&1
  echo "prof exit $?" >> gpurun_out/summary.txt
  find gpurun_out/prof -name "*kernel_stats*" | head -3 >> gpurun_out/summary.txt
  # keep the (large) raw trace out of the merge budget
  find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
fi
if [[ "$PHASES" == *dp2* ]]; then
  # the N > 1 code path of bench.py on ONE GPU: two ranks, gloo backend (rendezvous, broadcast, staged all-reduce, barriers,
  # max-over-ranks timing, rank-0 report); checks that it completes and prints one JSON line -- not a performance number
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 3 --warmup 1 --init-steps 2 --global-batch 16 --dist-backend gloo --no-cpu-baseline > gpurun_out/bench_dp2_gloo.log 2>&1
  echo "dp2 (gloo, shared GPU) exit $?" >> gpurun_out/summary.txt
  grep -c '"metric"' gpurun_out/bench_dp2_gloo.log >> gpurun_out/summary.txt
  tail -2 gpurun_out/bench_dp2_gloo.log | cut -c1-400 >> gpurun_out/summary.txt
fi
if [[ "$PHASES" == *p32* ]]; then
  # kernel trace at the per-GPU load of the 8-GPU run (32 images): sum of kernel durations vs wall = launch-gap share
  rm -rf gpurun_out/prof32
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof32" -o lbc -- python "$OLDPWD/bench.py" --dtype ${PROF_DTYPE:-bf16} --global-batch 32 --steps 20 --warmup 3 --init-steps 2 --no-cpu-baseline --no-alt) > gpurun_out/prof32.log 2>&1
  echo "prof32 exit $?" >> gpurun_out/summary.txt
  tail -1 gpurun_out/prof32.log | cut -c1-200 >> gpurun_out/summary.txt
  python scripts/trace_gaps.py $(find gpurun_out/prof32 -name "*kernel_trace.csv" | head -1) >> gpurun_out/summary.txt 2>&1
  find gpurun_out/prof32 -name "*kernel_trace*" -size +20M -delete
fi
if [[ "$PHASES" == *pmc* ]]; then
  # hardware counters: separate passes, kernel-trace only (no sys/runtime tracing together with --pmc)
  i=0
  for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf gpurun_out/pmc$i
    (cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc$i" -o lbc -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --init-steps 1 --no-cpu-baseline --no-alt) > gpurun_out/pmc$i.log 2>&1
    echo "pmc$i ($ctrs) exit $?" >> gpurun_out/summary.txt
    find gpurun_out/pmc$i -name "*kernel_trace*" -delete
    ls -la gpurun_out/pmc$i/* 2>/dev/null | head -5 >> gpurun_out/summary.txt
  done
fi
cat gpurun_out/summary.txt
