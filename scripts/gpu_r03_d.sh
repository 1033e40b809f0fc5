#!/bin/bash
# Round 3, GPU call 4: (1) ablations of the 256 x 256 shape of conv_hdma_k, (2) frozen-decision gradient parity at full size,
# (3) weight-gradient split count A/B on the step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for L in l3.conv l2.conv l4.conv; do
  for C in 0 1; do
    for D in 0 15 16 2 8 1; do
      echo "== $L cfg $C diag $D: $(LBC_HDMA_CFG=$C LBC_HDMA_DIAG=$D timeout 60 python scripts/bench_ops.py 256 3 fwd $L 2>&1 | grep "$L" | head -1)"
    done
  done
done 2>&1 | tee $R/hdma_diag_cfg0.log
rm -f $R/grad_diag.txt
timeout 900 python -m pytest tests/test_model.py -q -m gpu -k "frozen_decisions" -rP 2>&1 | tail -40 > $R/frozen_pytest.log
cat $R/grad_diag.txt
tail -5 $R/frozen_pytest.log
for B in 512 256 384; do
  echo "== wgrad_tr blocks $B: $(LBC_WGRAD_TR_BLOCKS=$B timeout 200 python bench.py --no-cpu-baseline --no-alt 2>&1 | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["by_kernel"]["conv_wgrad_tr"])')"
done 2>&1 | tee $R/wgrad_blocks.log
