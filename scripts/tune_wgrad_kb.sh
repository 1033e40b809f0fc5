for bm in 16384 4096; do echo "== BIGM=$bm"; LBC_WGRAD_BIGM=$bm PYTHONPATH=. timeout 300 python scripts/bench_ops.py ${1:-256} 12 wgrad 2>&1 | grep -v amdgpu.ids; done
