#!/bin/bash
# The 8-rank self-test of bench.py's N > 1 path on ONE GPU (eight processes share the device; gloo stages the buckets through the host, so
# the numbers mean nothing): the world size and the per-GPU load (32 images) of the metric's 8-GPU run -- rendezvous, rank-0 broadcast,
# six stage buckets per step (bf16 on the wire in the bf16 mode), max-over-ranks timing, ONE JSON line whose `comm_ranks` is read back from
# the buckets' communicator.  Then the same with --sync-bn, and 2 ranks on RCCL proper (both on device 0: RCCL may refuse -- reported).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
run() {
  local tag=$1 n=$2 be=$3 gb=$4
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29519 \
      bench.py --gpus $n --steps 3 --warmup 2 --init-steps 2 --global-batch $gb --dist-backend $be --pool-frames 128 --no-cpu-baseline --no-alt ${EXTRA:-} > $R/dp_$tag.log 2>&1
  echo "dp $tag exit $?: $(grep '"metric"' $R/dp_$tag.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms/step; world_size', d['world_size'],'; ranks read back from the communicator', d['comm_ranks'], '(', d['comm_backend'], '); loss_finite', d.get('loss_finite'), ';', d['config']['parallelism'])" 2>/dev/null)"
}
run gloo8_b256 8 gloo 256
EXTRA="--sync-bn" run gloo8_b256_syncbn 8 gloo 256
run rccl2_b64_one_device 2 nccl 64
