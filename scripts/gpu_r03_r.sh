#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for v in "LBC_DECODER_PASS_MIN_COUT=128" "LBC_DECODER_PASS_MIN_COUT=64" "LBC_DECODER_PASS_MIN_COUT=128" "LBC_DECODER_PASS_MIN_COUT=64"; do
  echo "$v: $(env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done 2>&1 | tee $R/decoder_pass64_ab.log
rm -f $R/launches_dec64.txt
LBC_DECODER_PASS_MIN_COUT=64 LBC_PROF_LAUNCHES=$R/launches_dec64.txt timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-alt --breakdown $R/breakdown_r.json 2>&1 | tail -1 | cut -c1-100
