"""Cost of the SyncBN hook on ONE GPU: a 1-rank RCCL group makes every all-reduce an identity, so the difference between
the two timings is what the hook itself adds per step (86+ callbacks from the native executor into torch.distributed, the
extra one-row reduces and the second finalize of each BatchNorm backward) -- not the xGMI latency of a real 8-rank run.
Both transports are timed: the library's own RCCL communicator (ncclAllReduce enqueued from C) and the torch.distributed callback.
usage: python scripts/syncbn_cost.py [per_gpu_batch] [steps]"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from learningbycheating_amd.training.native import NativeTrainer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(bench.free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    student, teacher = bench.build_models(dev, "image")
    for m in (student, teacher):
        m.precision = "bf16"
    tr = NativeTrainer(student, teacher, n, (3, 160, 384), dev, phase=1, lr=1e-4)
    g = torch.Generator().manual_seed(3)
    rgb = torch.randint(0, 256, (n, 160, 384, 3), generator=g, dtype=torch.uint8).to(dev)
    bv = (torch.rand((n, 192, 192, 7), generator=g) < 0.1).to(torch.uint8).mul_(255).to(dev)
    speed = (torch.rand(n, generator=g) * 10).to(dev)
    cmd = torch.zeros((n, 4), device=dev)
    cmd[torch.arange(n), torch.randint(0, 4, (n,), generator=g)] = 1

    def timed():
        for _ in range(8):
            tr.step(rgb, speed, cmd, birdview=bv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(rgb, speed, cmd, birdview=bv)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    local = timed()
    tr.eng.set_sync_bn(None)                                  # the library's RCCL communicator
    native = timed()
    tr.eng.set_sync_bn(dist.new_group(), native=False)        # callbacks into torch.distributed
    synced = timed()
    err = tr.eng._sync["error"]
    print({"per_gpu_batch": n, "steps": steps, "ms_per_step_local_bn": round(local, 3), "ms_per_step_sync_bn_rccl_native_1rank": round(native, 3),
           "ms_per_step_sync_bn_torch_callback_1rank": round(synced, 3), "callback_error": repr(err) if err else None})
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
