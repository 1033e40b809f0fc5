#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
timeout 600 python -m pytest tests/test_kernels.py -q -m gpu -k "hdma" -x 2>&1 | tail -2
for L in l2.conv l3.conv l4.conv; do
  for v in "LBC_NO_HDMA_PERSIST=1" "LBC_HDMAP_VAR=0" "LBC_HDMAP_VAR=2" "LBC_HDMAP_VAR=4" "LBC_HDMAP_VAR=6" "LBC_HDMAP_VAR=1"; do
    echo "== $L $v: $(env $v timeout 60 python scripts/bench_ops.py 256 3 fwd,dgrad $L 2>&1 | grep "$L" | tr '\n' ' ')"
  done
done 2>&1 | tee $R/hdmap_ops3.log
for v in 0 2 4; do
  echo "== profile var $v"; LBC_HDMAP_VAR=$v timeout 300 python scripts/hdmap_prof.py 256 l2.conv l3.conv 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $R/hdmap_prof3.log
timeout 200 python bench.py --no-cpu-baseline --no-alt 2>&1 | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("step", d["ms_per_step"], d["roofline"]["frac"], {k:(v["ms"],v["tflops"]) for k,v in d["roofline"]["by_kernel"].items() if "hdma" in k})'
