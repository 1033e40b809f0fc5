#!/bin/bash
# round-2 baseline: new bench.py (default + b32), kernel stats, PMC passes, dp2 (gloo, shared GPU) self-spawn + anomaly probes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python bench.py --breakdown $R/s4_breakdown_bf16.json > $R/s4_bench.log 2>&1; echo "bench exit $?"; tail -1 $R/s4_bench.log | cut -c1-1500
timeout 300 python bench.py --global-batch 32 --no-alt --no-cpu-baseline --breakdown $R/s4_breakdown_b32.json > $R/s4_bench_b32.log 2>&1; echo "b32 exit $?"; tail -1 $R/s4_bench_b32.log | cut -c1-400
timeout 300 python bench.py --resident --no-alt --no-cpu-baseline > $R/s4_bench_resident.log 2>&1; echo "resident exit $?"; tail -1 $R/s4_bench_resident.log | cut -c1-300
rm -rf $R/s4_prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$R/s4_prof" -o lbc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --init-steps 2 --pool-frames 512 --no-cpu-baseline --no-alt) > $R/s4_prof.log 2>&1; echo "prof exit $?"
find $R/s4_prof -name "*kernel_trace*" -size +20M -delete
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf $R/s4_pmc$i
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OLDPWD/$R/s4_pmc$i" -o lbc -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --init-steps 1 --pool-frames 512 --no-cpu-baseline --no-alt) > $R/s4_pmc$i.log 2>&1
  echo "pmc$i exit $?"; find $R/s4_pmc$i -name "*kernel_trace*" -delete
done
for v in NONE LBC_NO_HALO LBC_NO_SIDE_STREAM; do
  env $v=1 timeout 150 python bench.py --gpus 2 --dist-backend gloo --steps 2 --warmup 1 --init-steps 2 --global-batch 16 --pool-frames 64 --no-cpu-baseline --no-alt > $R/s4_dp2_$v.log 2>&1
  echo "dp2 $v exit $?: $(grep -o '"ms_per_step": [0-9.]*' $R/s4_dp2_$v.log | head -1) world $(grep -o '"world_size": [0-9]*' $R/s4_dp2_$v.log | head -1)"
done
