#!/bin/bash
# Evidence of the shipped code, ONE gpurun call (ROUND=r06 by default; the results are copied to profiles/${ROUND}_final_* afterwards):
# driver-default bench + breakdown, the per-GPU loads of 2 / 4 / 8 GPUs, BASELINE configs 2 / 4 / 5, the kernel-only (device-resident batch) rate,
# SERIALIZED rocprofv3 kernel stats (no side streams: per-kernel durations are those of kernels running alone), three PMC passes (MFMA busy,
# FETCH_SIZE, WRITE_SIZE) -> ${ROUND}_pmc_traffic.json, LDS bank-conflict pass, per-launch tables, kernel-trace gap analysis.
# PHASES (default "bench prof pmc tables"): add "tests" for the full GPU test suite + smoke in the same call.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out; ROUND=${ROUND:-r06}
PHASES="${PHASES:-bench prof pmc tables}"
S=$R/summary.txt; echo "== $(date) phases: $PHASES" > $S
if [[ "$PHASES" == *bench* ]]; then
  timeout 600 python bench.py --breakdown $R/breakdown_bs256_bf16.json > $R/bench_bf16.log 2>&1; echo "bench exit $?" >> $S; tail -1 $R/bench_bf16.log | cut -c1-300 >> $S
  for B in 128 64 32; do
    timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt --breakdown $R/breakdown_b${B}_bf16.json > $R/bench_b${B}_bf16.log 2>&1
    echo "b$B: $(tail -1 $R/bench_b${B}_bf16.log | cut -c1-200)" >> $S
  done
  timeout 300 python bench.py --h2d --no-cpu-baseline --no-alt > $R/bench_h2d_bf16.log 2>&1; echo "h2d (PCIe-inclusive): $(tail -1 $R/bench_h2d_bf16.log | cut -c1-200)" >> $S
  for WLD in phase1_bs64_fp32 birdview_bs128 phase2_bs128; do
    timeout 300 python bench.py --workload $WLD --steps 30 --warmup 5 --no-cpu-baseline --breakdown $R/breakdown_$WLD.json > $R/bench_$WLD.log 2>&1
    echo "$WLD: $(tail -1 $R/bench_$WLD.log | cut -c1-220)" >> $S
  done
fi
if [[ "$PHASES" == *bench* && -d scratch_prev ]]; then
  # boxes of the pool differ by up to 10 %: the previous round's tree (git archive of its last commit, built in scratch_prev/, not tracked)
  # on THIS box next to the shipped code, one device-resident batch re-fed every step (the one input mode both trees have)
  pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
  for B in 256 32; do
    for rep in 1 2; do
      (cd scratch_prev && timeout 300 python bench.py --resident --global-batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj) > $R/ab_prev_b${B}_$rep.txt 2>&1
      (timeout 300 python bench.py --resident --global-batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj) > $R/ab_head_b${B}_$rep.txt 2>&1
      echo "same box, $B images, run $rep: previous round's tree $(cat $R/ab_prev_b${B}_$rep.txt) | shipped code $(cat $R/ab_head_b${B}_$rep.txt)" >> $S
    done
  done
  # ... and the exact-f32 path (the one held to the 1e-3 waypoint bar), both trees, one run each
  echo "same box, 256 images, exact f32: previous round's tree $(cd scratch_prev && timeout 400 python bench.py --resident --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj) | shipped code $(timeout 400 python bench.py --resident --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
fi
if [[ "$PHASES" == *prof* ]]; then
  rm -rf $R/prof
  (cd /tmp && LBC_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof" -o lbc -- python "$OLDPWD/bench.py" --serial --steps 3 --warmup 1 --init-steps 2 --no-cpu-baseline --no-alt) > $R/prof.log 2>&1
  echo "prof exit $?" >> $S
  find $R/prof -name "*kernel_trace*" -size +20M -delete
fi
if [[ "$PHASES" == *pmc* ]]; then
  i=0
  for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1)); rm -rf $R/pmc$i
    (cd /tmp && LBC_NO_SIDE_STREAM=1 timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OLDPWD/$R/pmc$i" -o lbc -- python "$OLDPWD/bench.py" --serial --steps 1 --warmup 1 --init-steps 1 --no-cpu-baseline --no-alt) > $R/pmc$i.log 2>&1
    echo "pmc$i ($ctrs) exit $?" >> $S
    find $R/pmc$i -name "*kernel_trace*" -delete
  done
  P1=$(find $R/pmc1 -name "*counter_collection.csv" | head -1); P2=$(find $R/pmc2 -name "*counter_collection.csv" | head -1); P3=$(find $R/pmc3 -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py $P1 $P2 $P3 > $R/pmc_summary.txt 2>&1
  python scripts/pmc_traffic.py $P2 $P3 bf16 "profiles/${ROUND}_final_pmc_bf16/pass2.csv (FETCH_SIZE x 2, MI355X_MICROARCH.md gfx950 correction) + pass3.csv (WRITE_SIZE): rocprofv3 --pmc passes of \`LBC_NO_SIDE_STREAM=1 bench.py --serial --steps 1 --warmup 1 --init-steps 1 --no-cpu-baseline --no-alt\` (scripts/gpu_evidence.sh)" > $R/${ROUND}_pmc_traffic.json 2> $R/pmc_traffic_table.txt
  PYTHONPATH=scripts python scripts/pmc_lds.py $(find $R/pmc4 -name "*counter_collection.csv" | head -1) > $R/pmc_lds_conflicts.txt 2>&1
fi
if [[ "$PHASES" == *tables* ]]; then
  rm -f $R/launches_bs256.txt $R/launches_bs32.txt
  LBC_PROF_LAUNCHES=$R/launches_bs256.txt timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-alt --breakdown /dev/null > /dev/null 2>&1
  LBC_PROF_LAUNCHES=$R/launches_bs32.txt timeout 300 python bench.py --global-batch 32 --steps 3 --warmup 2 --no-cpu-baseline --no-alt --breakdown /dev/null > /dev/null 2>&1
  for B in 256 32; do
    rm -rf $R/trace
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$R/trace" -o lbc -- python "$OLDPWD/bench.py" --global-batch $B --steps 4 --warmup 2 --init-steps 2 --no-cpu-baseline --no-alt) > $R/trace.log 2>&1
    python scripts/trace_gaps.py $(find $R/trace -name "*kernel_trace.csv" | head -1) 2 > $R/trace_gaps_bs$B.txt 2>&1; head -1 $R/trace_gaps_bs$B.txt | cut -c1-260 >> $S
    rm -rf $R/trace
  done
  # free-running per-kernel durations at the metric's 8-GPU operating point (32 images per GPU)
  rm -rf $R/prof_b32
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/prof_b32" -o lbc -- python "$OLDPWD/bench.py" --global-batch 32 --steps 20 --warmup 5 --init-steps 2 --no-cpu-baseline --no-alt) > $R/prof_b32.log 2>&1
  cp $(find $R/prof_b32 -name "*kernel_stats.csv" | head -1) $R/kernel_stats_b32.csv 2>/dev/null; rm -rf $R/prof_b32
  timeout 200 python scripts/bench_ops.py 256 3 fwd,dgrad,wgrad > $R/per_shape_bs256.txt 2>&1
  timeout 120 python scripts/bench_ops.py 32 3 fwd,dgrad,wgrad > $R/per_shape_bs32.txt 2>&1
fi
if [[ "$PHASES" == *tests* ]]; then
  rm -f $R/grad_diag.txt
  timeout ${PYTEST_TIMEOUT:-1800} python -m pytest tests -m gpu -q --durations=8 > $R/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $S; tail -12 $R/pytest_gpu.log >> $S
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke.log 2>&1; echo "smoke exit $?: $(tail -1 $R/smoke.log)" >> $S
fi
cat $S
