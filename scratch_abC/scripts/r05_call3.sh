#!/bin/bash
# round 5, call 3: the cleaned-up library (18 switches, conv_hdma_k / conv_glds_k / PRE / schedule variants gone), the 128 x 128 eight-wave
# shape for layer 2 at 32 images, bn2's backward reduce fused into the next block's conv1 input gradient (conv_hdmap_k<.., EPI 4>):
# the full GPU test suite, then same-box A/Bs at 256 / 64 / 32 images.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r05 call3" > $S
timeout 900 python -m pytest tests -m gpu -q -x --durations=6 > $R/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $S; tail -12 $R/pytest_gpu.log >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
ab() {   # ab <batch> <label> <env...>
  local B=$1 L=$2; shift 2
  echo "b$B $L: $(env "$@" timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S
}
for rep in 1 2; do
  ab 256 "default" LBC_X=0
  ab 256 "bn2 reduce as its own pass (LBC_NO_BN_BWD_FUSE=2)" LBC_NO_BN_BWD_FUSE=2
done
for rep in 1 2; do
  ab 32 "default" LBC_X=0
  ab 32 "bn2 reduce as its own pass (LBC_NO_BN_BWD_FUSE=2)" LBC_NO_BN_BWD_FUSE=2
  ab 32 "layer 2 on 256 x 128 tiles as before (LBC_HDMA_SMALL_BELOW=0)" LBC_HDMA_SMALL_BELOW=0
done
ab 64 "default" LBC_X=0
ab 64 "bn2 reduce as its own pass (LBC_NO_BN_BWD_FUSE=2)" LBC_NO_BN_BWD_FUSE=2
ab 128 "default" LBC_X=0
ab 128 "bn2 reduce as its own pass (LBC_NO_BN_BWD_FUSE=2)" LBC_NO_BN_BWD_FUSE=2
cat $S
