#!/bin/bash
# copies what scripts/gpu_evidence.sh left in gpurun_out/ to profiles/${ROUND}_final_* (the tracked, judged copies); missing pieces are skipped
cd "$(dirname "$0")/.."; R=gpurun_out; P=profiles; ROUND=${ROUND:-r04}; F=$P/${ROUND}_final
c() { [ -e "$1" ] && cp "$1" "$2"; }
c $R/summary.txt ${F}_summary.txt; c $R/bench_bf16.log ${F}_bench_bf16.log; c $R/bench_h2d_bf16.log ${F}_bench_h2d_bf16.log
for B in 128 64 32; do c $R/bench_b${B}_bf16.log ${F}_bench_b${B}_bf16.log; c $R/breakdown_b${B}_bf16.json ${F}_breakdown_b${B}_bf16.json; done
for W in phase1_bs64_fp32 birdview_bs128 phase2_bs128; do c $R/bench_$W.log ${F}_bench_$W.log; c $R/breakdown_$W.json ${F}_breakdown_$W.json; done
c $R/breakdown_bs256_bf16.json ${F}_breakdown_bs256_bf16.json
K=$(find $R/prof -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$K" ] && cp $K ${F}_rocprofv3_kernel_stats_serial_bf16.csv
mkdir -p ${F}_pmc_bf16; i=0
for d in pmc1 pmc2 pmc3 pmc4; do i=$((i+1)); K=$(find $R/$d -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$K" ] && cp $K ${F}_pmc_bf16/pass$i.csv; done
c $R/pmc_summary.txt ${F}_pmc_summary_bf16.txt; c $R/${ROUND}_pmc_traffic.json $P/${ROUND}_pmc_traffic.json; c $R/pmc_traffic_table.txt ${F}_pmc_traffic_table.txt
c $R/pmc_lds_conflicts.txt ${F}_pmc_lds_conflicts.txt
c $R/per_shape_bs256.txt ${F}_per_shape_bs256.txt; c $R/per_shape_bs32.txt ${F}_per_shape_bs32.txt
c $R/launches_bs256.txt ${F}_per_launch_bs256.txt; c $R/launches_bs32.txt ${F}_per_launch_bs32.txt
c $R/trace_gaps_bs256.txt ${F}_trace_gaps_bs256.txt; c $R/trace_gaps_bs32.txt ${F}_trace_gaps_bs32.txt
c $R/pytest_gpu.log ${F}_pytest_gpu.log; c $R/smoke.log ${F}_smoke.log; c $R/grad_diag.txt ${F}_grad_diag.txt
ls $P | grep ${ROUND}_final | wc -l
