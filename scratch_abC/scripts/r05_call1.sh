#!/bin/bash
# round 5, call 1: land-or-kill LBC_HDMAP_PRE (BatchNorm-on-load inside the persistent convolution): its GPU parity tests, then
# same-box A/B at 256 and 32 images per GPU.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r05 call1: LBC_HDMAP_PRE tests + A/B" > $S
LBC_TEST_HDMAP_PRE=1 timeout 600 python -m pytest tests -m gpu -q -x -k "hdma_fwd_dgrad or bn1_on_load or hdmap-pre or hdmap_pre" > $R/pytest_hdmap_pre.log 2>&1; echo "pytest exit $?" >> $S; tail -5 $R/pytest_hdmap_pre.log >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
for B in 256 32; do
  for rep in 1 2; do
    for PRE in 0 1; do
      echo "b$B rep$rep LBC_HDMAP_PRE=$PRE: $(LBC_HDMAP_PRE=$PRE timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
    done
  done
done
LBC_HDMAP_PRE=1 timeout 300 python bench.py --global-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-alt --breakdown $R/breakdown_bs256_pre.json > $R/bench_bs256_pre.log 2>&1
cat $S
