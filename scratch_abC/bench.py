#!/usr/bin/env python
"""bench.py -- images/sec of the LbC training hot path on N MI355X.

Default workload = BASELINE.json's metric: images/sec ImagePolicyModelSS phase-1 train @ bs256, 1/2/4/8 MI355X.
One "step" = the reference's hot loop body (training/train_image_phase1.py:174-205):
    H2D of the next batch's uint8 frames (double-buffered, overlapped) -> teacher (BirdViewPolicyModelSS r18, eval) forward
    -> student (ImagePolicyModelSS r34, train) forward -> unprojection + L1 over 4 branches -> backward
    -> (RCCL gradient all-reduce) -> Adam
fed from a synthetic dataset of uint8 frames (>= 2048 frames per rank, walked batch by batch; the reference's LMDB loader hands out
the same frames as float32 CHW, 4x the bytes).  By default the whole dataset is RESIDENT IN HBM when the timed region starts (2048
frames = 0.9 GB of the 288 GB; every step reads a different batch of it, no PCIe in the timed region); --h2d keeps it in pinned host
memory and uploads every batch inside the timed region (double-buffered on a copy stream) -- the PCIe-inclusive rate quoted in
DESIGN.md.  Global batch is fixed at 256 for every N ("strong" scaling, as the metric is quoted): 256/N images per GPU.

Other workloads (--workload): phase1_bs64_fp32 (BASELINE config 2), birdview_bs128 (config 4, train_birdview.py:116-128),
phase2_bs128 (config 5, train_image_phase2.py:152-258 incl. the per-sample weight write-back to the host).

`python bench.py --gpus N` with N > 1 and no torch.distributed environment re-launches itself under
torch.distributed.run with N ranks (one per GPU, RCCL); under a launcher it just joins the job.

Also on the JSON line:
  roofline      -- the convolution family (MFMA bound): algorithmic FLOP / HIP-event time of every launch of one extra,
                   instrumented step (lbc_profile_* in the C ABI); traffic = HBM bytes per launch from the committed
                   rocprofv3 --pmc passes of this command (profiles/, FETCH_SIZE x 2 + WRITE_SIZE), when available
  roofline_hbm  -- the BatchNorm family (HBM bound): algorithmic bytes / HIP-event time
  cpu_baseline  -- the oracle's (torch-CPU restatement of the reference) step on the host cores (N = 1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA (same table)
PEAK_HBM_GBS = 8000.0

# algorithmic FLOP per image-step: student fwd 9.441 + bwd 18.593 + teacher fwd 3.178 GFLOP; bird-view r18 9.129
WORKLOADS = {
    "phase1_bs256_bf16": dict(metric="images/sec ImagePolicyModelSS phase-1 train @ bs256", batch=256, dtype="bf16", kind="phase1", gflop=31.212,
                              what="ImagePolicyModelSS(resnet34) phase-1 step vs BirdViewPolicyModelSS(resnet18) teacher"),
    "phase1_bs64_fp32": dict(metric="images/sec ImagePolicyModelSS phase-1 train @ bs64 fp32 (BASELINE config 2)", batch=64, dtype="f32", kind="phase1", gflop=31.212,
                             what="ImagePolicyModelSS(resnet34) phase-1 step vs BirdViewPolicyModelSS(resnet18) teacher"),
    "birdview_bs128": dict(metric="images/sec BirdViewPolicyModelSS train_birdview @ bs128 (BASELINE config 4)", batch=128, dtype="f32", kind="birdview", gflop=9.129,
                           what="BirdViewPolicyModelSS(resnet18) behaviour-cloning step (L1 vs ground-truth waypoints)"),
    "phase2_bs128": dict(metric="images/sec ImagePolicyModelSS phase-2 train @ bs128 (BASELINE config 5)", batch=128, dtype="f32", kind="phase2", gflop=31.212,
                         what="phase-2 replay step: teacher + student forward, phase-1 loss, backward, Adam, per-sample weight write-back"),
}
DTYPE_TEXT = {"bf16_mfma": "bf16 MFMA operands + f32 tensors/accumulate/master/BN/loss/Adam",
              "bf16": "bf16 MFMA operands and bf16 activation storage + f32 accumulate/master weights/gradients/BN/loss/Adam",
              "f32": "exact-f32 MFMA"}


class DevicePool:
    """The synthetic dataset resident in HBM (the default): the same tensors as FramePool, on the device; a step takes the next
    `batch` frames as views -- no copy of any kind in the timed region."""

    def __init__(self, host):
        from learningbycheating_amd.bird_view.utils.train_utils import one_hot
        dev = host.device
        self.n, self.batch, self.device, self.pos = host.n, host.batch, dev, 0
        self.t = {"bv": host.bv.to(dev), "speed": host.speed.to(dev), "loc": host.loc.to(dev), "onehot": one_hot(host.cmd).to(dev)}
        if host.rgb is not None:
            self.t["rgb"] = host.rgb.to(dev)
        self.bytes_per_step = 0
        self.cur = None

    def prefetch(self, k):
        pass

    def get(self, k):
        if self.pos + self.batch > self.n:
            self.pos = 0
        s = slice(self.pos, self.pos + self.batch)
        self.pos = (self.pos + self.batch) % self.n
        return {key: v[s] for key, v in self.t.items()}

    def release(self, k):
        pass


class FramePool:
    """The synthetic dataset: uint8 frames as the reference's LMDB files hold them (rgb HWC; the 7 bird-view channels as
    0/255 masks, cropped to 192x192), speed, command and ground-truth waypoints, in PINNED host memory, plus two device
    slots filled by asynchronous H2D copies on a side stream while the previous step computes."""

    def __init__(self, n_frames, batch, device, seed, need_rgb=True, slots=True):
        n_frames = max(n_frames, 2 * batch)
        n_frames = (n_frames + batch - 1) // batch * batch
        g = torch.Generator().manual_seed(seed)
        self.n, self.batch, self.device = n_frames, batch, device
        self.rgb = None
        if need_rgb:
            self.rgb = torch.empty((n_frames, 160, 384, 3), dtype=torch.uint8).pin_memory()
            for s in range(0, n_frames, 256):
                e = min(n_frames, s + 256)
                self.rgb[s:e] = torch.randint(0, 256, (e - s, 160, 384, 3), generator=g, dtype=torch.uint8)
        self.bv = torch.empty((n_frames, 192, 192, 7), dtype=torch.uint8).pin_memory()
        for s in range(0, n_frames, 256):
            e = min(n_frames, s + 256)
            self.bv[s:e] = (torch.rand((e - s, 192, 192, 7), generator=g) < 0.1).to(torch.uint8) * 255
        self.speed = (torch.rand(n_frames, generator=g) * 10).pin_memory()
        self.cmd = torch.randint(1, 5, (n_frames,), generator=g).float()
        self.loc = (torch.rand((n_frames, 5, 2), generator=g) * 192).pin_memory()

        def slot():
            d = {"bv": torch.empty((batch, 192, 192, 7), dtype=torch.uint8, device=device), "speed": torch.empty(batch, device=device),
                 "onehot": torch.empty((batch, 4), device=device), "loc": torch.empty((batch, 5, 2), device=device)}
            if need_rgb:
                d["rgb"] = torch.empty((batch, 160, 384, 3), dtype=torch.uint8, device=device)
            return d
        self.slots = [slot(), slot()] if slots else None
        self.copy = torch.cuda.Stream(device=device)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [torch.cuda.Event(), torch.cuda.Event()]
        for e in self.free:
            e.record()
        self.pos = 0
        self.bytes_per_step = batch * ((160 * 384 * 3 if need_rgb else 0) + 192 * 192 * 7 + 4 + 16 + 40)

    def prefetch(self, k):
        """enqueue the H2D copies of the next batch into slot k (call right after the step that used slot k was enqueued)"""
        from learningbycheating_amd.bird_view.utils.train_utils import one_hot
        self.ready[k].synchronize()          # the host never runs more than two steps ahead
        s = slice(self.pos, self.pos + self.batch)
        self.pos = (self.pos + self.batch) % self.n
        # reference bird_view/utils/train_utils.py:33-40, per batch, on the host.  The 4 KB result goes up from pageable memory:
        # a pinned buffer that the GPU has read is expensive to rewrite on this platform (inference.py, scripts/diag_latency.py);
        # the frame arrays below are written once and only ever read by the DMA engine
        onehot = one_hot(self.cmd[s])
        d = self.slots[k]
        with torch.cuda.stream(self.copy):
            self.copy.wait_event(self.free[k])               # the step that read slot k has finished with it
            if self.rgb is not None:
                d["rgb"].copy_(self.rgb[s], non_blocking=True)
            d["bv"].copy_(self.bv[s], non_blocking=True)
            d["speed"].copy_(self.speed[s], non_blocking=True)
            d["loc"].copy_(self.loc[s], non_blocking=True)
            d["onehot"].copy_(onehot, non_blocking=True)
            self.ready[k].record(self.copy)

    def get(self, k):
        torch.cuda.current_stream(self.device).wait_event(self.ready[k])
        return self.slots[k]

    def release(self, k):
        self.free[k].record(torch.cuda.current_stream(self.device))


def build_models(device, kind, seed=0):
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS
    student = teacher = None
    if kind != "birdview":
        torch.manual_seed(seed)
        student = ImagePolicyModelSS("resnet34", all_branch=True).to(device)
    torch.manual_seed(seed + 1)
    teacher = BirdViewPolicyModelSS("resnet18", all_branch=(kind != "birdview")).to(device)
    return student, teacher


def cpu_baseline(kind):
    """the oracle's step on the host cores at batch 8 (BASELINE config 1's batch: `value`) and at batch 64 (BASELINE.md section 3 asks
    for both), each a bounded sample; `cores` = the intra-op threads used, `host_cores` = what the box has"""
    small = cpu_baseline_at(kind, 8, 14.0, 40)
    large = cpu_baseline_at(kind, 64, 16.0, 5)
    small["host_cores"] = os.cpu_count()
    small["batch64"] = {k: large[k] for k in ("value", "unit", "sample")}
    return small


def cpu_baseline_at(kind, batch, seconds_budget, max_steps):
    """oracle (port of the reference step onto torch-CPU functional ops) timed on the host cores"""
    from oracle import lbc_oracle as O
    # intra-op threads: all cores up to 64 (a 23 M-parameter CNN at batch 8 stops scaling, and slows down, far below
    # the 256 hardware threads of the GPU host); `cores` reports the threads actually used
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(7)
    rgb = (torch.randint(0, 256, (batch, 160, 384, 3), generator=g, dtype=torch.uint8).permute(0, 3, 1, 2).float() / 255.0).contiguous()
    bv = (torch.rand((batch, 7, 192, 192), generator=g) < 0.1).float()
    speed = torch.rand(batch, generator=g) * 10
    onehot = O.one_hot(torch.randint(1, 5, (batch,), generator=g).float())
    if kind == "birdview":
        ssd = O.as_params(O.make_state_dict("birdview", "resnet18", 2, trained_like=False))
        loc = torch.rand((batch, 5, 2), generator=g) * 192
        params = [v for v in ssd.values() if v.requires_grad]

        def loss_fn():
            return O.birdview_loss(O.policy_forward(ssd, "birdview", "resnet18", bv, speed, onehot, True)[0], loc)
        what = "oracle train_birdview step (r18 fwd/bwd + Adam)"
    else:
        ssd = O.as_params(O.make_state_dict("image", "resnet34", 1, trained_like=False))
        tsd = O.make_state_dict("birdview", "resnet18", 2)
        params = [v for v in ssd.values() if v.requires_grad]

        def loss_fn():
            return O.phase1_step_loss(ssd, tsd, "resnet34", "resnet18", rgb, bv, speed, onehot)[0]
        what = "oracle phase-1 step (teacher r18 fwd + student r34 fwd/bwd + Adam)"
    opt = torch.optim.Adam(params, lr=1e-4)

    def step():
        opt.zero_grad()
        loss_fn().mean().backward()
        opt.step()

    step()
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < seconds_budget and n < max_steps):
        step()
        n += 1
    dt = time.time() - t0
    return {"value": round(batch * n / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%s, batch %d x %d steps, torch %s CPU" % (what, batch, n, torch.__version__)}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_spawn(args):
    """plain `python bench.py --gpus N`: run the N ranks under torch.distributed.run, relay their output, return the exit code"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def read_traffic():
    """HBM bytes per launch of the convolution family from the committed PMC passes of THIS round's code: profiles/r06_pmc_traffic.json,
    written by scripts/pmc_traffic.py from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_evidence.sh (counters need
    their own runs: they cannot be collected inside this process); an older round's file is a fallback and says so in its `source`"""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            try:
                return json.load(open(p))
            except Exception:
                pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 50 timed steps after >= 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="phase1_bs256_bf16")
    ap.add_argument("--global-batch", type=int, default=None, help="default: the workload's batch")
    ap.add_argument("--dtype", choices=["f32", "bf16_mfma", "bf16"], default=None,
                    help="default: the workload's.  f32: exact-f32 MFMA everywhere (the parity path). bf16: mixed precision of BASELINE.json "
                         "config 3 -- bf16 MFMA operands and bf16 activation storage, f32 accumulation / master weights / gradients / "
                         "BatchNorm / soft-argmax / loss / Adam. bf16_mfma: bf16 MFMA operands only, every tensor f32")
    ap.add_argument("--grad-allreduce", choices=["f32", "bf16", "auto"], default="auto",
                    help="dtype of the gradient buckets on the wire (N > 1): auto = bf16 in the bf16 mode (BASELINE config 3), f32 otherwise")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N > 1: BatchNorm over the global batch (lbc_net_set_sync_bn; 2 small all-reduces per BatchNorm per step on a "
                         "communicator of their own). Default: local statistics per rank, like torch DDP without SyncBatchNorm")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL over xGMI, one rank per GPU); gloo lets several ranks share one GPU to exercise the N > 1 code "
                         "path on a single-GPU box -- its numbers mean nothing")
    ap.add_argument("--pool-frames", type=int, default=2048, help="frames per rank in the synthetic dataset")
    ap.add_argument("--h2d", action="store_true", help="keep the dataset in pinned host memory and upload every batch inside the timed region "
                                                       "(double-buffered): the PCIe-inclusive rate.  Default: the dataset is resident in HBM")
    ap.add_argument("--resident", action="store_true", help="re-feed ONE device-resident batch every step")
    ap.add_argument("--init-steps", type=int, default=40, help="below-horizon warm start (stands in for the phase-0 checkpoint)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the short extra runs of the default line: exact f32 ('also'), PCIe-inclusive "
                                                          "('h2d_inclusive') and the 32-image per-GPU load of the 8-GPU run ('per_gpu_32')")
    ap.add_argument("--breakdown", default=None, help="write the per-kernel-class profile of the instrumented step here (json)")
    ap.add_argument("--serial", action="store_true",
                    help="profiling runs: the teacher forward stays on the main stream (with LBC_NO_SIDE_STREAM=1 every kernel then runs alone, "
                         "so that rocprofv3 per-kernel durations are not inflated by co-running kernels)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    kind = wl["kind"]
    global_batch = args.global_batch or wl["batch"]
    dtype = args.dtype or wl["dtype"]

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_spawn(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    local = local % torch.cuda.device_count()      # (several ranks share a device only in the gloo self-test)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus

    from learningbycheating_amd import _lib, WAYPOINT_TOLERANCE
    from learningbycheating_amd.parallel import broadcast_module
    from learningbycheating_amd.training.native import NativeTrainer
    assert _lib.backend() == "hip-gfx950"
    lib = _lib.get()

    per_gpu = global_batch // world
    assert per_gpu * world == global_batch, "global batch must divide by the number of GPUs"
    host_pool = FramePool(per_gpu if args.resident else args.pool_frames, per_gpu, device, 1000 + rank, need_rgb=kind != "birdview", slots=True)
    # the default feed: everything the timed region reads lives in HBM (the pinned host copy only serves --h2d and the short
    # PCIe-inclusive run the default line reports next to the headline number)
    dev_pool = None if (args.h2d or args.resident) else DevicePool(host_pool)
    cur = {"pool": host_pool if dev_pool is None else dev_pool, "pg": per_gpu}
    # Warm start below the horizon: the phase-1 unprojection has a 1/y pole at the horizon and the reference always
    # starts phase 1 from a phase-0 checkpoint (train_image_phase1.py:244); a few L1 steps towards below-horizon targets
    # stand in for it.  Not timed.
    g = torch.Generator().manual_seed(5 + rank)
    tgt = torch.rand((per_gpu, 4, 5, 2), generator=g)
    tgt[..., 0] = tgt[..., 0] * 1.2 - 0.6
    tgt[..., 1] = tgt[..., 1] * 0.5 + 0.3
    tgt = tgt.to(device)

    def run_steps(tr, n, mode, weights=None):
        """n steps through the double-buffered H2D pipeline (or on one resident batch); returns the last loss tensor"""
        loss = None
        pool = cur["pool"]
        if args.resident:
            b = pool.get(0)
        for i in range(n):
            k = i & 1
            if not args.resident:
                b = pool.get(k)
            if mode == "warm":
                loss = tr.step(b["rgb"], b["speed"], b["onehot"], target=tgt[:cur["pg"]])
            elif kind == "birdview":
                loss = tr.step(b["bv"], b["speed"], b["onehot"], target=b["loc"])
            else:
                loss = tr.step(b["rgb"], b["speed"], b["onehot"], birdview=b["bv"])
                if kind == "phase2":
                    # train_image_phase2.py:203-206: per-sample resampling weights go back to the (host-side) replay buffer
                    from learningbycheating_amd.training.train_image_phase2 import phase2_weights
                    weights.append(phase2_weights(tr, tr.last_pred[0], tr.last_teacher[0]).cpu())
            if not args.resident:
                pool.release(k)
                pool.prefetch(k)
        return loss

    def timed_run(dt_name, steps, warmup, pool=None, pg=None):
        """pool / pg: another feed (the pinned host dataset) or another per-GPU batch for this run; default = the run's own"""
        if pool is not None:
            cur["pool"] = pool
        pool = cur["pool"]
        cur["pg"] = pg or per_gpu
        pool.batch = cur["pg"]
        pg = cur["pg"]
        student, teacher = build_models(device, kind)
        prec = {"f32": "fp32", "bf16_mfma": "bf16_mfma", "bf16": "bf16"}[dt_name]
        for m in (student, teacher):
            if m is not None:
                m.precision = prec
                broadcast_module(m)
        pool.pos = 0
        pool.prefetch(0); pool.prefetch(1)
        gdt = torch.bfloat16 if (args.grad_allreduce == "bf16" or (args.grad_allreduce == "auto" and dt_name == "bf16")) else None
        if kind == "birdview":
            tr = NativeTrainer(teacher, None, pg, (7, 192, 192), device, phase="birdview", lr=1e-4, world_size=world, grad_dtype=gdt, sync_bn=args.sync_bn)
        else:
            warm = NativeTrainer(student, None, pg, (3, 160, 384), device, phase="l1_all", lr=1e-3, world_size=world, grad_dtype=gdt)
            run_steps(warm, args.init_steps, "warm")
            del warm
            tr = NativeTrainer(student, teacher, pg, (3, 160, 384), device, phase=1, lr=1e-4, world_size=world, grad_dtype=gdt,
                               sync_bn=args.sync_bn)
        if args.serial:
            tr.overlap_teacher = False
        weights = []
        run_steps(tr, warmup, "train", weights)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = run_steps(tr, steps, "train", weights)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tr, float(t.item()), float(loss.mean().item())

    def instrumented_step(tr, dt_name):
        """one extra step with HIP events around every launch (every rank runs it -- it contains the all-reduce -- rank 0 reports)"""
        if rank == 0:
            lib.lbc_profile_enable(1)
        tr.overlap_teacher = False       # one stream: the events of this step bracket kernels that run alone
        run_steps(tr, 1, "train", [])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if rank != 0:
            return None, None, None
        lib.lbc_profile_enable(0)
        import ctypes
        buf = ctypes.create_string_buffer(1 << 16)
        nbytes = lib.lbc_profile_report(buf, len(buf))
        br = {}
        for line in buf.raw[:nbytes].decode().strip().splitlines():
            name, cnt, ms, fl, by = line.split()
            br[name] = {"launches": int(cnt), "ms": float(ms), "gflop": float(fl) / 1e9, "gbyte": float(by) / 1e9}
        total_ms = sum(v["ms"] for v in br.values())
        conv = {k: v for k, v in br.items() if k.startswith("conv_")}
        ms = sum(v["ms"] for v in conv.values()); gf = sum(v["gflop"] for v in conv.values()); n = sum(v["launches"] for v in conv.values())
        peak = PEAK_BF16_MFMA_TFLOPS if dt_name != "f32" else PEAK_FP32_MFMA_TFLOPS
        mf = "v_mfma_f32_32x32x16_bf16" if dt_name != "f32" else "v_mfma_f32_32x32x2_f32"
        ach = gf / ms if ms > 0 else 0.0     # GFLOP / ms = TFLOP/s
        roof = {"bound": "mfma", "kernel": "convolution family (%s): %s" % (mf, ", ".join(sorted(conv))), "achieved": round(ach, 2),
                "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "instrumented_step": "serialized: one extra step on one stream with HIP events around every launch (the timed steps overlap the "
                                     "teacher forward -- and, in the f32 modes, the weight gradients -- on side streams, and carry no per-launch "
                                     "events: the sum of these durations exceeds ms_per_step)",
                "launches_per_step": n, "avg_launch_ms": round(ms / max(n, 1), 4), "gflop_per_launch": round(gf / max(n, 1), 3),
                "share_of_step_kernel_time": round(ms / total_ms, 3) if total_ms else None,
                "by_kernel": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "tflops": round(v["gflop"] / v["ms"], 1) if v["ms"] else None}
                              for k, v in sorted(conv.items())}}
        tj = read_traffic()
        if tj and dt_name in tj:
            t = tj[dt_name]        # {"traffic_bytes_per_launch", "algorithmic_bytes_per_launch", "kernel", "source"}
            roof["traffic"] = t.get("traffic_bytes_per_launch")
            gb = sum(v["gbyte"] for v in conv.values())
            roof["algorithmic_bytes_per_launch"] = round(gb * 1e9 / max(n, 1))
            roof["traffic_over_algorithmic"] = round(roof["traffic"] * max(n, 1) / (gb * 1e9), 3) if gb and roof["traffic"] else None
            # round 4's definition of the algorithmic bytes (the side tensors of the fused BatchNorm-backward launches not counted): kept next to
            # the current one so that the ratio stays comparable across rounds
            side = br.get("side_tensors_of_fused_reduce", {}).get("gbyte", 0.0)
            roof["algorithmic_bytes_per_launch_without_fused_side_tensors"] = round((gb - side) * 1e9 / max(n, 1))
            roof["traffic_over_algorithmic_without_fused_side_tensors"] = round(roof["traffic"] * max(n, 1) / ((gb - side) * 1e9), 3) if gb - side > 0 and roof["traffic"] else None
            roof["traffic_kernel"] = t.get("kernel")
            roof["traffic_source"] = t.get("source")
        bn = {k: v for k, v in br.items() if k in ("bn_apply", "bn_bwd_reduce", "bn_bwd_apply", "channel_stats")}
        bms = sum(v["ms"] for v in bn.values()); bgb = sum(v["gbyte"] for v in bn.values())
        hbm = {"bound": "hbm", "kernel": "BatchNorm family: " + ", ".join(sorted(bn)), "achieved": round(bgb / bms * 1e3, 1) if bms else 0.0,
               "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(bgb / bms * 1e3 / PEAK_HBM_GBS, 4) if bms else 0.0,
               "launches_per_step": sum(v["launches"] for v in bn.values()), "ms_per_step": round(bms, 3),
               "non_conv_kernel_ms_per_step": round(total_ms - ms, 3)}
        return roof, hbm, br

    tr, dt, loss_mean = timed_run(dtype, args.steps, args.warmup)
    # the ranks the gradient buckets really travel between: a one from every rank summed on the buckets' own communicator and stream
    comm_ranks = tr.reducer.participants() if world > 1 else None        # read back from the buckets' communicator, whatever its backend
    roof, hbm, breakdown = instrumented_step(tr, dtype)
    if rank == 0 and args.breakdown:
        os.makedirs(os.path.dirname(os.path.abspath(args.breakdown)), exist_ok=True)
        with open(args.breakdown, "w") as f:
            json.dump({"workload": args.workload, "dtype": dtype, "per_gpu_batch": per_gpu, "step_ms_timed": 1e3 * dt / args.steps, "classes": breakdown}, f, indent=1)

    out = None
    if rank == 0:
        value = global_batch * args.steps / dt
        feed = ("ONE device-resident batch re-fed every step" if args.resident else
                ("%d-frame pinned host dataset per rank, uint8 H2D of every batch (%.1f MB) double-buffered inside the timed region" % (host_pool.n, host_pool.bytes_per_step / 1e6)
                 if args.h2d else "%d-frame dataset per rank resident in HBM, a different batch every step" % host_pool.n))
        out = {"metric": wl["metric"], "value": round(value, 2), "unit": "images/sec",
               "n_gpus": world, "world_size": dist.get_world_size() if world > 1 else 1,
               # ranks of the communicator the gradient buckets actually travelled on, read back from it (StageAllReducer.participants;
               # None: one process, nothing travels); `comm_backend` says whether that communicator is RCCL
               "comm_ranks": comm_ranks, "comm_backend": (("rccl" if args.dist_backend == "nccl" else args.dist_backend) if world > 1 else None),
               "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": "%s [%s], 160x384 RGB + 7x192x192 bird-view uint8 NHWC frames as the dataset stores them (%s), global batch %d "
                                      "(%d/GPU), %s, %s BatchNorm, %sAdam lr 1e-4" % (wl["what"], args.workload, feed, global_batch, per_gpu, DTYPE_TEXT[dtype],
                                         "synchronized (global-batch)" if (args.sync_bn and world > 1) else "local",
                                         "" if world == 1 else ("%s gradient buckets over %s, " % ("bf16" if (args.grad_allreduce == "bf16" or (args.grad_allreduce == "auto" and dtype == "bf16")) else "f32", "RCCL" if args.dist_backend == "nccl" else args.dist_backend))),
                          "global_batch": global_batch, "parallelism": "dp%d" % world,
                          "waypoint_tolerance_vs_f32": WAYPOINT_TOLERANCE[{"f32": "fp32"}.get(dtype, dtype)]},
               "loss": loss_mean, "loss_finite": bool(loss_mean == loss_mean and abs(loss_mean) != float("inf")),
               "algorithmic_tflops": round(value * wl["gflop"] / 1e3, 2),
               "roofline": roof, "roofline_hbm": hbm}
    also = None
    if dtype != "f32" and not args.no_alt:
        # the exact-f32 parity path on the same workload, a short run reported next to the headline number
        del tr
        torch.cuda.empty_cache()
        asteps = max(3, min(20, args.steps // 2))
        atr, adt, aloss = timed_run("f32", asteps, 2)
        aroof, ahbm, _ = instrumented_step(atr, "f32")
        also = {"dtype": "f32", "value": round(global_batch * asteps / adt, 2), "ms_per_step": round(1e3 * adt / asteps, 3),
                "steps": asteps, "note": "exact-f32 MFMA path (the one held to the 1e-3 waypoint parity bar)", "roofline": aroof, "roofline_hbm": ahbm}
    h2d_incl = small = None
    if world == 1 and not args.no_alt and not args.h2d and not args.resident and dtype != "f32":
        # (a) SURVEY 8(d) writes the metric with the uint8 H2D of every batch inside the timed region: the same step fed from the pinned host
        #     dataset (113 MB per 256-image batch, double-buffered on a copy stream) -- never `value`, reported next to it
        torch.cuda.empty_cache()
        hsteps = max(5, min(20, args.steps // 2))
        host_pool.pos = 0
        htr, hdt, _ = timed_run(dtype, hsteps, 5, pool=host_pool)
        h2d_incl = {"ms_per_step": round(1e3 * hdt / hsteps, 3), "value": round(global_batch * hsteps / hdt, 2), "steps": hsteps,
                    "h2d_mb_per_step": round(host_pool.bytes_per_step / 1e6, 1),
                    "note": "the same step with the uint8 frames uploaded from pinned host memory inside the timed region (double-buffered): the PCIe-inclusive rate"}
        del htr
        # (b) the metric's 8-GPU operating point is 32 images per GPU: that load on this one GPU (no communication), with its own roofline
        if kind == "phase1" and global_batch == 256:
            torch.cuda.empty_cache()
            str_, sdt, _ = timed_run(dtype, 30, 10, pool=dev_pool, pg=32)
            sroof, shbm, _ = instrumented_step(str_, dtype)
            small = {"per_gpu_batch": 32, "ms_per_step": round(1e3 * sdt / 30, 3), "value_one_gpu": round(32 * 30 / sdt, 2), "steps": 30,
                     "no_comm_projection_8gpu": round(8 * 32 * 30 / sdt, 1),
                     "note": "the per-GPU load of the 8-GPU run (256 / 8 images) on ONE GPU, no communication: what the 8-GPU value can at most be is 8 x value_one_gpu",
                     "roofline": {k: sroof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches_per_step", "avg_launch_ms")},
                     "roofline_hbm": {k: shbm[k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches_per_step", "ms_per_step")}}
            del str_
    if rank == 0:
        out["also"] = also
        out["h2d_inclusive"] = h2d_incl
        out["per_gpu_32"] = small
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0 must not keep the other ranks waiting)
            out["cpu_baseline"] = cpu_baseline(kind)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
