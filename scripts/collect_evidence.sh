#!/bin/bash
# copies what scripts/gpu_r03_evidence.sh left in gpurun_out/ to profiles/r03_final_* (the tracked, judged copies)
cd "$(dirname "$0")/.."; R=gpurun_out; P=profiles
cp $R/summary.txt $P/r03_final_summary.txt; cp $R/bench_bf16.log $P/r03_final_bench_bf16.log
for B in 128 64 32; do cp $R/bench_b${B}_bf16.log $P/r03_final_bench_b${B}_bf16.log; cp $R/breakdown_b${B}_bf16.json $P/r03_final_breakdown_b${B}_bf16.json; done
cp $R/breakdown_bs256_bf16.json $P/r03_final_breakdown_bs256_bf16.json
cp $(find $R/prof -name "*kernel_stats.csv" | head -1) $P/r03_final_rocprofv3_kernel_stats_serial_bf16.csv
mkdir -p $P/r03_final_pmc_bf16; i=0
for d in pmc1 pmc2 pmc3; do i=$((i+1)); cp $(find $R/$d -name "*counter_collection.csv" | head -1) $P/r03_final_pmc_bf16/pass$i.csv; done
cp $R/pmc_summary.txt $P/r03_final_pmc_summary_bf16.txt; cp $R/r03_pmc_traffic.json $P/r03_pmc_traffic.json; cp $R/pmc_traffic_table.txt $P/r03_final_pmc_traffic_table.txt
cp $R/per_shape_bs256.txt $P/r03_final_per_shape_bs256.txt; cp $R/per_shape_bs32.txt $P/r03_final_per_shape_bs32.txt
cp $R/launches_bs256.txt $P/r03_final_per_launch_bs256.txt; cp $R/launches_bs32.txt $P/r03_final_per_launch_bs32.txt
cp $R/trace_gaps_bs256.txt $P/r03_final_trace_gaps_bs256.txt; cp $R/trace_gaps_bs32.txt $P/r03_final_trace_gaps_bs32.txt
cp $R/wgrad_group_bs256.txt $P/r03_final_wgrad_group_bs256.txt
cp $R/pytest_gpu.log $P/r03_final_pytest_gpu.log; cp $R/smoke.log $P/r03_final_smoke.log; cp $R/grad_diag.txt $P/r03_final_grad_diag.txt
