for tb in 256 512 1024; do echo "== BLOCKS=$tb"; LBC_WGRAD_BLOCKS=$tb PYTHONPATH=. timeout 300 python scripts/bench_ops.py ${1:-256} 2 wgrad 2>&1 | grep -v amdgpu.ids; done
