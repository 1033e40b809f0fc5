#!/bin/bash
# the 2-rank gloo self-test of bench.py on ONE GPU (two processes share the device): which kernels make the bf16 step pathological there?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
run() {
  local tag=$1; shift
  env "$@" HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 2 --warmup 1 --init-steps 1 --global-batch 16 --dist-backend gloo --no-cpu-baseline --no-alt > $R/dp2_$tag.log 2>&1
  echo "dp2 $tag exit $?: $(grep '"metric"' $R/dp2_$tag.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms/step', d['world_size'],'ranks', d['config']['workload'][-90:])" 2>/dev/null)"
}
run default LBC_DUMMY=0
run nohalo_nowgradtr LBC_NO_HALO=1 LBC_NO_WGRAD_TR=1
run nosidestream LBC_NO_SIDE_STREAM=1
