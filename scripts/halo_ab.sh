cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k halo 2>&1 | tail -3
echo "== halo on"; PYTHONPATH=. python scripts/bench_ops.py ${1:-256} 3 fwd,fwd+bn,dgrad 2>&1 | grep -v amdgpu | grep conv
