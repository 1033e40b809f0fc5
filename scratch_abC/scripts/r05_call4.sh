#!/bin/bash
# round 5, call 4: the in-workgroup K split of the four-wave persistent convolution (conv_hdmap_k<.., KG = 2>): its GPU parity tests, then
# same-box A/B against LBC_HDMAP_SPLIT=0 (no K split of any kind) at 32 / 64 / 256 images
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r05 call4" > $S
timeout 900 python -m pytest tests -m gpu -q -x -k "hdma_fwd_dgrad or bn_backward_reduce_fused or frozen_decisions or launch_policy or k_steps_bf16 or split" > $R/pytest_gpu_kg2.log 2>&1; echo "pytest exit $?" >> $S; tail -6 $R/pytest_gpu_kg2.log >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
ab() { local B=$1 L=$2; shift 2; echo "b$B $L: $(env "$@" timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S; }
for rep in 1 2 3; do
  ab 32 "default (K split inside the workgroup for launches of <= 256 four-wave tiles)" LBC_X=0
  ab 32 "LBC_HDMAP_SPLIT=0 (no K split)" LBC_HDMAP_SPLIT=0
done
for rep in 1 2; do
  ab 64 "default" LBC_X=0
  ab 64 "LBC_HDMAP_SPLIT=0" LBC_HDMAP_SPLIT=0
done
ab 16 "default" LBC_X=0
ab 16 "LBC_HDMAP_SPLIT=0" LBC_HDMAP_SPLIT=0
ab 256 "default" LBC_X=0
timeout 120 python scripts/bench_ops.py 32 3 fwd,dgrad > $R/per_shape_bs32_kg2.txt 2>&1
LBC_HDMAP_SPLIT=0 timeout 120 python scripts/bench_ops.py 32 3 fwd,dgrad > $R/per_shape_bs32_nosplit.txt 2>&1
cat $S
