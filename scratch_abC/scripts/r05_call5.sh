#!/bin/bash
# round 5, call 5: (a) the tightened bf16 accuracy tests with their diagnostics (ratio to autocast on three batches, per-group frozen-decision
# errors, 12-seed fit) -> the measured values the bounds are set from; (b) sweep of the folded-finalize parameters at 32 / 64 images
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
S=$R/summary.txt; echo "== $(date) r05 call5" > $S; rm -f $R/grad_diag.txt
timeout 900 python -m pytest tests/test_model.py -m gpu -q -k "bf16_mode_declared_accuracy or bf16_gradients_with_frozen or bf16_phase1_fit or bf16_gradients_match_autocast" > $R/pytest_gpu_bf16.log 2>&1; echo "pytest exit $?" >> $S; tail -25 $R/pytest_gpu_bf16.log >> $S
cat $R/grad_diag.txt >> $S
pj() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')" 2>&1 | tail -1; }
ab() { local B=$1 L=$2; shift 2; echo "b$B $L: $(env "$@" timeout 300 python bench.py --global-batch $B --steps 50 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | pj)" >> $S; }
for B in 32 64; do
  ab $B "default (fold grid 512, 128 KB of rows)" LBC_X=0
  ab $B "fold grid 256" LBC_TUNE_FOLD_GRID=256
  ab $B "fold grid 1024" LBC_TUNE_FOLD_GRID=1024
  ab $B "fold rows 64 KB" LBC_TUNE_FOLD_KB=64
  ab $B "fold rows 256 KB" LBC_TUNE_FOLD_KB=256
  ab $B "no fold (LBC_NO_BN_FOLD=1)" LBC_NO_BN_FOLD=1
  ab $B "default again" LBC_X=0
done
cat $S
