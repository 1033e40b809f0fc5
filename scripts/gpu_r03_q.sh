#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
rm -f $R/launches_bs256.txt
LBC_PROF_LAUNCHES=$R/launches_bs256.txt timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --breakdown $R/breakdown_q.json 2>&1 | tail -1 | cut -c1-200
wc -l $R/launches_bs256.txt
