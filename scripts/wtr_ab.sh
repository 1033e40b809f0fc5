cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels.py -m gpu -q -k "tap_fused or wgrad" 2>&1 | tail -3
echo "== tr on"; PYTHONPATH=. python scripts/bench_ops.py ${1:-256} 2 wgrad,wgrad+bn 2>&1 | grep -v amdgpu | grep -E "conv|c1"
echo "== tr off"; LBC_NO_WGRAD_TR=1 PYTHONPATH=. python scripts/bench_ops.py ${1:-256} 2 wgrad 2>&1 | grep -v amdgpu | grep -E "conv|c1"
