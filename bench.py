#!/usr/bin/env python
"""bench.py -- images/sec of the ImagePolicyModelSS phase-1 training step on N MI355X.

Metric (BASELINE.json): images/sec ImagePolicyModelSS phase-1 train @ bs256, 1/2/4/8 MI355X.
One "step" = the reference's hot loop body (training/train_image_phase1.py:174-205):
teacher (BirdViewPolicyModelSS r18, eval) forward -> student (ImagePolicyModelSS r34, train)
forward -> unprojection + L1 over 4 branches -> backward -> (RCCL gradient all-reduce) -> Adam,
on synthetic frames already resident in HBM.  Global batch is fixed at 256 for every N
("strong" scaling, as the metric is quoted): 256/N images per GPU.

Also reported on the same JSON line:
  roofline      -- the dominant kernel class (the fp32 MFMA implicit-GEMM convolutions), from HIP-event
                   timing of every launch of one extra, instrumented step (lbc_profile_* in the C ABI)
  cpu_baseline  -- the oracle's (torch-CPU restatement of the reference) phase-1 step on the host cores
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMAGE_STEP = 31.212e9      # student fwd 9.441 + bwd 18.593 + teacher fwd 3.178 GFLOP
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA (same table)
PEAK_HBM_GBS = 8000.0


def synthetic_batch(n, device, seed, uint8_frames=False):
    """frames as the reference's LMDB dataset holds them (uint8 HWC; bird-view 7 binary channels stored as 0/255) and, by
    default, decoded the way its loader hands them to the model (float32 CHW in [0,1]); uint8_frames keeps them uint8 NHWC
    for the executor's fused input pass (lbc_net_forward_u8)"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    rgb = torch.randint(0, 256, (n, 160, 384, 3), generator=g, dtype=torch.uint8)
    bv = ((torch.rand((n, 192, 192, 7), generator=g) < 0.1).to(torch.uint8) * 255)
    speed = torch.rand(n, generator=g) * 10
    cmd = torch.randint(1, 5, (n,), generator=g).float()
    if not uint8_frames:
        rgb = (rgb.permute(0, 3, 1, 2).float() / 255.0).contiguous()
        bv = (bv.permute(0, 3, 1, 2).float() / 255.0).contiguous()
    return rgb.to(device), bv.to(device), speed.to(device), cmd


def build_models(device, seed=0):
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS
    torch.manual_seed(seed)
    student = ImagePolicyModelSS("resnet34", all_branch=True).to(device)
    torch.manual_seed(seed + 1)
    teacher = BirdViewPolicyModelSS("resnet18", all_branch=True).to(device)
    return student, teacher


def cpu_baseline(seconds_budget=25.0, batch=8):
    """oracle (port of the reference step onto torch-CPU functional ops) timed on the host cores"""
    from oracle import lbc_oracle as O
    # intra-op threads: all cores up to 64 (a 23 M-parameter CNN at batch 8 stops scaling, and slows down, far below
    # the 256 hardware threads of the GPU host); `cores` reports the threads actually used
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    ssd = O.as_params(O.make_state_dict("image", "resnet34", 1, trained_like=False))
    tsd = O.make_state_dict("birdview", "resnet18", 2)
    params = [v for v in ssd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    rgb, bv, speed, cmd = synthetic_batch(batch, "cpu", 7)
    onehot = O.one_hot(cmd)

    def step():
        loss, _, _, _ = O.phase1_step_loss(ssd, tsd, "resnet34", "resnet18", rgb, bv, speed, onehot)
        opt.zero_grad()
        loss.mean().backward()
        opt.step()

    step()
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < seconds_budget and n < 40):
        step()
        n += 1
    dt = time.time() - t0
    return {"value": round(batch * n / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "oracle phase-1 step (teacher r18 fwd + student r34 fwd/bwd + Adam), batch %d x %d steps, torch %s CPU" % (batch, n, torch.__version__)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--global-batch", type=int, default=256)
    ap.add_argument("--dtype", choices=["f32", "bf16_mfma", "bf16"], default="bf16",
                    help="f32: exact-f32 MFMA everywhere (the parity path). bf16: mixed precision of BASELINE.json config 3 -- bf16 MFMA "
                         "operands and bf16 activation storage, f32 accumulation / master weights / gradients / BatchNorm / soft-argmax / "
                         "loss / Adam. bf16_mfma: bf16 MFMA operands only, every tensor f32")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL over xGMI, one rank per GPU); gloo lets two ranks share one GPU to exercise the N > 1 code path "
                         "on a single-GPU box (scripts/gpu_round.sh phase dp2) -- its numbers mean nothing")
    ap.add_argument("--float-input", action="store_true",
                    help="feed float32 NCHW frames (the reference loader's output) instead of the dataset's uint8 NHWC frames")
    ap.add_argument("--init-steps", type=int, default=40, help="below-horizon warm start (stands in for the phase-0 checkpoint)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the short exact-f32 run reported under 'also'")
    ap.add_argument("--breakdown", default=None, help="write the per-kernel-class profile of the instrumented step here (json)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    local = local % torch.cuda.device_count()      # (several ranks may share a device only in the gloo self-test below)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    from learningbycheating_amd import _lib
    from learningbycheating_amd.parallel import broadcast_module
    from learningbycheating_amd.training.native import NativeTrainer
    assert _lib.backend() == "hip-gfx950"

    per_gpu = args.global_batch // world
    assert per_gpu * world == args.global_batch, "global batch must divide by the number of GPUs"
    rgb, bv, speed, cmd = synthetic_batch(per_gpu, device, 1000 + rank, uint8_frames=not args.float_input)
    from learningbycheating_amd.bird_view.utils.train_utils import one_hot
    onehot = one_hot(cmd).to(device)
    # Warm start below the horizon: the phase-1 unprojection has a 1/y pole at the horizon and the reference always
    # starts phase 1 from a phase-0 checkpoint (train_image_phase1.py:244); a few L1 steps towards below-horizon targets
    # stand in for it.  Not timed.
    g = torch.Generator().manual_seed(5 + rank)
    tgt = torch.rand((per_gpu, 4, 5, 2), generator=g)
    tgt[..., 0] = tgt[..., 0] * 1.2 - 0.6
    tgt[..., 1] = tgt[..., 1] * 0.5 + 0.3
    tgt = tgt.to(device)

    def timed_run(dtype, steps, warmup):
        student, teacher = build_models(device)
        student.precision = teacher.precision = {"f32": "fp32", "bf16_mfma": "bf16_mfma", "bf16": "bf16"}[dtype]
        broadcast_module(student); broadcast_module(teacher)
        warm = NativeTrainer(student, None, per_gpu, (3, 160, 384), device, phase="l1_all", lr=1e-3, world_size=world)
        for _ in range(args.init_steps):
            warm.step(rgb, speed, onehot, target=tgt)
        del warm
        tr = NativeTrainer(student, teacher, per_gpu, (3, 160, 384), device, phase=1, lr=1e-4, world_size=world)
        for _ in range(warmup):
            loss = tr.step(rgb, speed, onehot, birdview=bv)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.step(rgb, speed, onehot, birdview=bv)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tr, float(t.item()), float(loss.mean().item())

    tr, dt, loss_mean = timed_run(args.dtype, args.steps, args.warmup)

    # ---- one extra instrumented step: HIP events around every kernel launch -------------------------
    # Every rank runs the step (it contains the gradient all-reduce: a rank-0-only step would deadlock N > 1); only rank 0
    # switches the per-launch HIP events on and reports.
    roof, breakdown = None, None
    import ctypes
    lib = _lib.get()
    lib.lbc_profile_enable.restype = ctypes.c_int
    lib.lbc_profile_report.restype = ctypes.c_int
    if rank == 0:
        lib.lbc_profile_enable(1)
    tr.overlap_teacher = False       # one stream: the HIP events of this step bracket kernels that run alone
    tr.step(rgb, speed, onehot, birdview=bv)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if rank == 0:
        lib.lbc_profile_enable(0)
        buf = ctypes.create_string_buffer(1 << 16)
        nbytes = lib.lbc_profile_report(buf, len(buf))
        breakdown = {}
        for line in buf.raw[:nbytes].decode().strip().splitlines():
            name, cnt, ms, fl, by = line.split()
            breakdown[name] = {"launches": int(cnt), "ms": float(ms), "gflop": float(fl) / 1e9, "gbyte": float(by) / 1e9}
        conv = [v for k, v in breakdown.items() if k.startswith("conv_igemm") or k == "conv_wgrad"]
        ms = sum(v["ms"] for v in conv); gf = sum(v["gflop"] for v in conv); n = sum(v["launches"] for v in conv)
        total_ms = sum(v["ms"] for v in breakdown.values())
        ach = gf / ms if ms > 0 else 0.0     # GFLOP / ms = TFLOP/s
        peak = PEAK_BF16_MFMA_TFLOPS if args.dtype != "f32" else PEAK_FP32_MFMA_TFLOPS
        kname = "conv_igemm_k / conv_wgrad_bf16_k (v_mfma_f32_32x32x16_bf16)" if args.dtype != "f32" else "conv_igemm_k / conv_wgrad_f32 (v_mfma_f32_32x32x2_f32)"
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2),
                "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "launches_per_step": n, "avg_launch_ms": round(ms / max(n, 1), 4), "gflop_per_launch": round(gf / max(n, 1), 3),
                "share_of_step_kernel_time": round(ms / total_ms, 3) if total_ms else None}
        if args.breakdown:
            os.makedirs(os.path.dirname(os.path.abspath(args.breakdown)), exist_ok=True)
            with open(args.breakdown, "w") as f:
                json.dump({"per_gpu_batch": per_gpu, "step_ms_timed": 1e3 * dt / args.steps, "classes": breakdown}, f, indent=1)

    if rank == 0:
        value = args.global_batch * args.steps / dt
        out = {"metric": "images/sec ImagePolicyModelSS phase-1 train @ bs256", "value": round(value, 2), "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "ImagePolicyModelSS(resnet34) phase-1 step vs BirdViewPolicyModelSS(resnet18) teacher, "
                                      "160x384 RGB + 7x192x192 bird-view (%s, resident in HBM), global batch %d (%d/GPU), %s, local BatchNorm, "
                                      "Adam lr 1e-4" % ("float32 NCHW frames" if args.float_input else "uint8 NHWC frames as the dataset stores them", args.global_batch, per_gpu, {"bf16_mfma": "bf16 MFMA operands + f32 tensors/accumulate/master/BN/loss/Adam", "bf16": "bf16 MFMA operands and bf16 activation storage + f32 accumulate/master weights/gradients/BN/loss/Adam", "f32": "exact-f32 MFMA"}[args.dtype]),
                          "global_batch": args.global_batch, "parallelism": "dp%d" % world},
               "loss": loss_mean, "loss_finite": bool(loss_mean == loss_mean and abs(loss_mean) != float("inf")),
               "algorithmic_tflops": round(value * FLOP_PER_IMAGE_STEP / 1e12, 2),
               "roofline": roof}
        out["_alt"] = None
    also = None
    if args.dtype != "f32" and not args.no_alt:
        # the exact-f32 parity path on the same workload, a short run reported next to the headline number
        del tr
        torch.cuda.empty_cache()
        asteps = max(3, args.steps // 2)
        _, adt, aloss = timed_run("f32", asteps, 2)
        also = {"dtype": "f32", "value": round(args.global_batch * asteps / adt, 2), "ms_per_step": round(1e3 * adt / asteps, 3),
                "steps": asteps, "note": "exact-f32 MFMA path (the one held to the 1e-3 waypoint parity bar)"}
    if rank == 0:
        out.pop("_alt")
        out["also"] = also
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0 must not keep the other ranks waiting)
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
