cd $GRAFT_REPO_ROOT
PYTHONPATH=. python scripts/bench_ops.py 256 3 fwd,dgrad 2>&1 | grep -v amdgpu
PYTHONPATH=. python scripts/bench_ops.py 32 3 fwd,dgrad 2>&1 | grep -v amdgpu | grep -E "conv"
