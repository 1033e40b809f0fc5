# tile-policy sweep of the generic convolution (LBC_FORCE_CFG 0 = 128x64, 1 = 128x128, 2 = 64x64) and split target of the
# tap-fused weight gradient, at the two batch sizes that matter (256 = one GPU, 32 = per-GPU load of the 8-GPU run)
cd ${GRAFT_REPO_ROOT:-.}
for b in 256 32; do
  for c in 0 1 2; do echo "== batch $b cfg $c"; LBC_FORCE_CFG=$c PYTHONPATH=. python scripts/bench_ops.py $b 3 fwd,dgrad 2>&1 | grep -E "conv|c1"; done
  for t in 256 512 1024; do echo "== batch $b wgrad_tr_blocks $t"; LBC_WGRAD_TR_BLOCKS=$t PYTHONPATH=. python scripts/bench_ops.py $b 2 wgrad 2>&1 | grep -E "conv"; done
done
