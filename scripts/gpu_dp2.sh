#!/bin/bash
# the 2-rank self-test of bench.py's N > 1 path on ONE GPU (two processes share the device): rendezvous, broadcast, staged backward with the
# per-stage gradient buckets behind the deferred grouped weight gradients, max-over-ranks timing, one JSON line.  gloo (host-staged buckets)
# and nccl (RCCL, both ranks on device 0 -- may be refused by RCCL: reported, not fatal)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
run() {
  local tag=$1 be=$2 gb=$3
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 3 --warmup 2 --init-steps 2 --global-batch $gb --dist-backend $be --no-cpu-baseline --no-alt ${EXTRA:-} > $R/dp2_$tag.log 2>&1
  echo "dp2 $tag exit $?: $(grep '"metric"' $R/dp2_$tag.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms/step', d['world_size'],'ranks', 'loss_finite', d.get('loss_finite'))" 2>/dev/null)"
}
run gloo_b16 gloo 16
run gloo_b64 gloo 64
# the same with synchronized BatchNorm (torch.distributed callback transport under gloo; rows fenced behind the buckets)
EXTRA="--sync-bn" run gloo_b16_syncbn gloo 16
timeout 300 python -m pytest tests/test_parallel.py -q -m gpu 2>&1 | tail -2
