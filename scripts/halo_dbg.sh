cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 3 7; do echo "== dbg $d"; LBC_HALO_DBG=$d PYTHONPATH=. python scripts/bench_ops.py 256 3 fwd 2>&1 | grep "l1.conv"; done
