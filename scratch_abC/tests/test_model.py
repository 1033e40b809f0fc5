"""Whole-network parity: the HIP executor against the oracle (torch-CPU restatement) and against the committed
outputs of the REAL reference classes (tests/golden).  Unmarked cases run the kernel sources under the CPU
emulator on reduced image sizes; gpu-marked cases run the gfx950 library at the reference's sizes."""
import ctypes
import os

import numpy as np
import pytest
import torch

from learningbycheating_amd import _lib
from oracle import lbc_oracle as O
from oracle.make_golden import seeded_inputs
from tests.helpers import engine_from_state_dict, relerr

gpu = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _inputs(kind, n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((n, 3, h, w), generator=g) if kind == "image" else (torch.rand((n, 7, h, w), generator=g) < 0.2).float()
    speed = torch.rand(n, generator=g) * 10
    cmd = O.one_hot(torch.randint(1, 5, (n,), generator=g).float())
    return x, speed, cmd


def _diag(dev, msg):
    """evidence lines (forward / gradient error distributions): printed (pytest -rP shows them) and, on the GPU box, appended to
    gpurun_out/grad_diag.txt, which is copied to profiles/ with the round's other logs"""
    print(msg)
    if torch.device(dev).type == "cuda":
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "grad_diag.txt"), "a") as f:
            f.write(msg + "\n")


def _fwd_bwd_check(dev, kind, backbone, h, w, n, fwd_tol, grad_tol, calibrated=True, flip_tol=2e-4, truth64=True):
    """truth64: ground truth = the oracle in float64 (default); False = the float32 oracle (large batches: the float64
    autograd graph of a ResNet-34 at batch 64 needs tens of GB of host memory)"""
    tdt = torch.float64 if truth64 else torch.float32
    sd = O.make_state_dict(kind, backbone, 3, h, w)
    x, speed, cmd = _inputs(kind, n, h, w, 4)
    if calibrated:
        O.calibrate_running_stats(sd, kind, backbone, x, speed, cmd)
    eng, tens = engine_from_state_dict(sd, kind, backbone, h, w, n, dev)
    xd, sdv, cd = x.to(dev), speed.to(dev), cmd.to(dev)
    # float64 oracle = ground truth; the float32 oracle's own distance to it measures the conditioning of the case
    # (eval mode with synthetic running statistics lets activations grow to ~5e4, where fp32 round-off alone moves
    # the soft-argmax by ~3e-4).  The HIP result must be within max(fwd_tol, 4x that distance) of the truth.
    sd64 = {k: (v.to(tdt) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    fwd_err = {}

    def check_forward(train, ps, pa, sd32):
        with torch.no_grad():
            o32s, o32a = O.policy_forward(sd32, kind, backbone, x, speed, cmd, train)
            o64s, o64a = (o32s, o32a) if not truth64 else O.policy_forward({k: v.clone() for k, v in sd64.items()}, kind, backbone, x.double(), speed.double(), cmd.double(), train)
        cond = (o32a.double() - o64a.double()).abs().max().item()
        tol = max(fwd_tol, 4 * cond)
        e = max((pa.cpu().double() - o64a.double()).abs().max().item(), (ps.cpu().double() - o64s.double()).abs().max().item())
        fwd_err[train] = (e, cond)
        assert e < tol, ("forward train=%s" % train, e, tol, cond)
        assert e < 1e-3, "north-star bar"

    # eval mode (running statistics)
    ps, pa = eng.forward(xd, sdv, cd, False)
    check_forward(False, ps, pa, {k: v.clone() for k, v in sd.items()})
    # training mode: batch statistics, running-stat update, backward
    ps, pa = eng.forward(xd, sdv, cd, True)
    sp = {k: v.clone() for k, v in sd.items()}
    check_forward(True, ps, pa, sp)
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(tens[k].cpu(), sp[k], rtol=1e-4, atol=1e-5), k
        if k.endswith("num_batches_tracked"):
            assert tens[k].item() == 1, k
    g = torch.Generator().manual_seed(5)
    d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g), torch.randn((n, 5, 2), generator=g)
    eng.backward(d_sel.to(dev), d_all.to(dev))
    # Gradient ground truth = the oracle in float64.  The network is piecewise (ReLU masks, max-pool arg-max), so any
    # two float32 evaluations (torch-CPU vs HIP) occasionally take different branches for an element that sits at a
    # kink; such a flip moves a few gradient tensors by a finite amount without either side being wrong.  Hence:
    # the typical (median) error must be at round-off level, the 90th percentile within grad_tol, and nothing gross.
    sp64 = O.as_params(sd64)
    ops64, opa64 = O.policy_forward(sp64, kind, backbone, x.to(tdt), speed.to(tdt), cmd.to(tdt), True)
    ((opa64 * d_all.to(tdt)).sum() + (ops64 * d_sel.to(tdt)).sum()).backward()
    # the float32 oracle's own error against float64 calibrates what "round-off + occasional branch flip" means here
    if truth64:
        sp32 = O.as_params(sd)
        o32s, o32a = O.policy_forward(sp32, kind, backbone, x, speed, cmd, True)
        ((o32a * d_all).sum() + (o32s * d_sel).sum()).backward()
    else:
        sp32 = sp64
    errs, noise = [], []
    names = list(eng.grad_views.keys())
    for k in names:
        v, ref = eng.grad_views[k], sp64[k].grad
        if k.startswith("location_pred") and k.endswith(".1.bias"):
            # a per-channel bias cancels inside the softmax: the true gradient is 0 up to round-off
            assert v.abs().max().item() < 1e-5
            continue
        if k.startswith("location_pred") and k.endswith(".0.bias"):
            assert (v.cpu().double() - ref).abs().max().item() < 1e-5 + grad_tol * ref.abs().max().item()
            continue
        errs.append((relerr(v.cpu().double(), ref.double()), k))
        noise.append(relerr(sp32[k].grad.double(), ref.double()))
    # How tight can this be?  A weight/bias gradient is a sum of ~1e5-1e6 random-sign terms, so ONE element whose
    # pre-activation sits within float32 round-off of a ReLU kink (or a max-pool tie) and flips between two float32
    # evaluations moves rel-to-max entries by ~1/sqrt(n) ~ 1e-3..1e-2, and contaminates everything upstream of it.  The
    # float32 oracle shows the same effect against float64 (measured on the GPU box: up to 5e-2 in layer2).  Therefore:
    # (1) head parameters (first in the backward pass, no ReLU above them): tight bound
    for e, k in errs:
        if k.startswith("location_pred"):
            assert e < grad_tol, (k, e)
    # (2) whole network: flip-tolerant bounds; a wrong kernel gives O(1) errors in many tensors
    es, ns = sorted(e for e, _ in errs), sorted(noise)
    med, p90 = es[len(es) // 2], es[int(len(es) * 0.9)]
    _diag(dev, "engine f32 %s %s %dx%d N=%d (%s truth): |pred - oracle| eval %.2e (f32-oracle's own %.2e) train %.2e (%.2e); "
               "gradients rel-to-max over %d tensors: median %.2e p90 %.2e max %.2e (%s); f32 oracle vs truth: median %.2e p90 %.2e max %.2e"
          % (kind, backbone, h, w, n, "float64" if truth64 else "float32", fwd_err[False][0], fwd_err[False][1], fwd_err[True][0], fwd_err[True][1],
             len(es), med, p90, es[-1], sorted(errs)[-1][1], ns[len(ns) // 2], ns[int(len(ns) * 0.9)], ns[-1]))
    if truth64:
        assert med < flip_tol, ("median gradient error", med, "float32-oracle median", ns[len(ns) // 2])
        assert p90 < 3 * flip_tol, ("90th percentile gradient error", p90, "float32-oracle p90", ns[int(len(ns) * 0.9)])
        assert es[-1] < max(0.2, 3 * ns[-1]), ("gross gradient error", sorted(errs)[-1], ns[-1])
    else:
        # two float32 evaluations (HIP vs torch-CPU) each carry their own kink flips: the rel-to-max entries are ~2x those of one
        # evaluation against float64 (measured at batch 64: median 4.5e-2, max 0.26 on an entry of deconv.7.weight while that
        # tensor's cosine is 0.9989).  Direction is the robust statistic here: every tensor within 0.99 cosine, typical 0.999;
        # a wrong kernel term (tests/test_ops.py, tests/test_kernels.py check each tightly) moves whole tensors, not entries.
        cs = sorted(torch.nn.functional.cosine_similarity(eng.grad_views[k].cpu().double().reshape(1, -1), sp64[k].grad.double().reshape(1, -1)).item()
                    for _, k in errs)
        _diag(dev, "   per-tensor gradient cosines vs the float32 oracle: min %.5f p10 %.5f median %.5f" % (cs[0], cs[len(cs) // 10], cs[len(cs) // 2]))
        assert cs[0] > 0.99 and cs[len(cs) // 2] > 0.998, cs[:5]
        assert med < 2.5 * flip_tol and p90 < 6 * flip_tol and es[-1] < 0.5, (med, p90, es[-1])
    # (3) per-tensor gradient norms agree (insensitive to single flips)
    for k in names:
        if k.startswith("location_pred") and k.endswith("bias"):
            continue          # analytically zero (both the 1x1 conv's bias and the BatchNorm's beta cancel in the softmax): round-off only
        a, b = eng.grad_views[k].cpu().double().norm().item(), sp64[k].grad.double().norm().item()
        if b > 1e-6:
            assert abs(a - b) <= 0.05 * b, ("gradient norm", k, a, b)
    return es[-1]


def frozen_decisions(eng):
    """The branch decisions of the executor's last training-mode forward, read back from its workspace (lbc_net_activation_info):
    {site: 0/1 mask in NCHW} for every ReLU, the positive mask and chosen window tap of the fused relu + max-pool -- the
    `frozen` argument of oracle.policy_forward."""
    acts = eng.activations()
    to = lambda t: t.permute(0, 3, 1, 2).cpu()
    fz = {"conv.maxpool": to(acts["conv.maxpool"]).float() > 0, "conv.maxpool.idx": to(acts["conv.maxpool.idx"])}
    for name, t in acts.items():
        if name.endswith(".conv1") and name != "conv.conv1":
            p = name[:-len(".conv1")]
            sc, sh = acts[p + ".bn1.scale"].reshape(1, -1, 1, 1).cpu(), acts[p + ".bn1.shift"].reshape(1, -1, 1, 1).cpu()
            # the executor evaluates relu(y1 * scale + shift) on the STORED y1 in float32 -- on gfx950 as one fused multiply-add
            # (in float64 the product of two floats is exact, so the sign of the float64 sum is the sign of the fma), under the CPU
            # emulator (x86-64 without FMA contraction) as a rounded product plus a rounded sum, like torch's float32 ops.  One
            # element whose |z| ~ 1e-9 decides differently between the two moves layer-1 bias gradients by 7e-3 at test sizes.
            y1 = to(t).float()
            fz[p + ".bn1"] = ((y1.double() * sc.double() + sh.double()) > 0) if eng.workspace.device.type == "cuda" else ((y1 * sc + sh) > 0)
            fz[p] = to(acts[p]).float() > 0
        elif name.startswith("deconv."):
            fz[name] = to(t).float() > 0
    return fz


def _frozen_gradient_check(dev, kind, backbone, h, w, n, precision, tol, head_tol=None, seed=3, sd=None):
    """Gradient parity with the branch decisions frozen.  The float64 oracle is run with the ReLU masks and max-pool choices
    the executor's own forward took (frozen_decisions), so both sides differentiate the SAME piecewise-linear function and the
    comparison is not blurred by kink flips (two float32 evaluations of a 34-layer ReLU network take different branches for a
    few of ~1e8 elements; each flip moves rel-to-max gradient entries by 1e-3..1e-2).  A mis-scaled term in any of the
    BatchNorm / convolution / pooling backward kernels shows up as an O(1e-2..1) error in the tensors upstream of it.
    Returns the sorted per-tensor errors (max |g - g64| / max |g64|)."""
    x, speed, cmd = _inputs(kind, n, h, w, seed + 1)
    if sd is None:                       # (sd given: a trained-like checkpoint the caller prepared)
        sd = O.make_state_dict(kind, backbone, seed, h, w)
        O.calibrate_running_stats(sd, kind, backbone, x, speed, cmd)
    eng, tens = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=precision)
    ps, pa = eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), True)
    fz = frozen_decisions(eng)           # (before the backward pass reuses any buffer)
    g = torch.Generator().manual_seed(seed + 2)
    d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g), torch.randn((n, 5, 2), generator=g)
    eng.backward(d_sel.to(dev), d_all.to(dev))
    sp = O.as_params({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()})
    # precision 2: the float64 oracle also rounds where the executor rounds (MFMA operands and stored activations / activation
    # gradients to bf16, oracle flags MFMA_BF16 / ACT_BF16) -- same decisions, same rounding points, float64 in between
    O.MFMA_BF16 = O.ACT_BF16 = (precision == 2)
    try:
        ops, opa = O.policy_forward(sp, kind, backbone, x.double(), speed.double(), cmd.double(), True, frozen=fz)
        ((opa * d_all.double()).sum() + (ops * d_sel.double()).sum()).backward()
    finally:
        O.MFMA_BF16 = O.ACT_BF16 = False
    fwd = max((pa.cpu().double() - opa.detach()).abs().max().item(), (ps.cpu().double() - ops.detach()).abs().max().item())
    errs = []
    for k, v in eng.grad_views.items():
        ref = sp[k].grad
        if k.startswith("location_pred") and k.endswith("bias"):
            # analytically zero (a per-channel offset cancels in the softmax): absolute check
            assert v.abs().max().item() < 1e-4 * max(1.0, float(d_all.abs().max())), (k, v.abs().max().item())
            continue
        errs.append((relerr(v.cpu().double(), ref), k))
    errs.sort()
    es = [e for e, _ in errs]
    group = lambda k: "head" if k.startswith("location_pred") else ("decoder" if k.startswith("deconv") else ("stem+layer1" if k.startswith(("conv.conv1", "conv.bn1", "conv.layer1")) else "layers2-4"))
    gmax = {}
    for e, k in errs:
        gmax[group(k)] = max(gmax.get(group(k), 0.0), e)
    _diag(dev, "frozen-decision gradient check, precision %d %s %s %dx%d N=%d: forward |pred - frozen float64 oracle| %.2e; gradients "
               "rel-to-max over %d tensors: median %.2e p90 %.2e max %.2e (%s); worst per group: %s"
          % (precision, kind, backbone, h, w, n, fwd, len(es), es[len(es) // 2], es[int(len(es) * 0.9)], es[-1], errs[-1][1],
             ", ".join("%s %.2e" % (g, v) for g, v in sorted(gmax.items()))))
    for e, k in errs:
        # tol: one bound, or one per tensor group {"head", "decoder", "layers2-4", "stem+layer1"}
        t = tol[group(k)] if isinstance(tol, dict) else (head_tol if (head_tol is not None and k.startswith("location_pred")) else tol)
        assert e < t, ("frozen-decision gradient", k, e, t)
    return es


@pytest.mark.parametrize("precision,tol", [(0, 1e-4), (2, 0.35)])
@pytest.mark.parametrize("kind,backbone,h,w,n", [("birdview", "resnet18", 64, 64, 4), ("image", "resnet18", 32, 64, 5)])
def test_gradients_with_frozen_decisions_emulated(env, kind, backbone, h, w, n, precision, tol):
    """exact-f32 path: every gradient within 1e-4 rel-to-max of the float64 oracle (measured 2.5e-5).  bf16 path at these sizes
    (2 x 2 maps in layer 4, BatchNorm over 16 values, an untrained network): two bf16 evaluations with the same rounding points
    already differ by 7e-2 in the waypoints (torch's own bf16 autocast of the oracle: 8e-2), so the bound only catches wiring
    errors here; the full-size bound is asserted on the GPU."""
    dev, _ = env
    _frozen_gradient_check(dev, kind, backbone, h, w, n, precision, tol)


@gpu
@pytest.mark.parametrize("kind,backbone,h,w,n", [("image", "resnet34", 160, 384, 4), ("image", "resnet34", 160, 384, 32),
                                                 ("birdview", "resnet18", 192, 192, 4)])
def test_gradients_with_frozen_decisions_full_size(env, kind, backbone, h, w, n):
    """all parameter gradients of the reference-sized networks on the exact-f32 path vs the float64 oracle on the executor's own
    branch decisions: <= 3e-4 rel-to-max per tensor, median <= 1e-4 (measured on MI355X: r34 N = 4 / 32 median 6.3e-5 / 6.7e-5, max
    1.1e-4 / 1.7e-4 on a head weight; r18 bird-view max 4.1e-5 -- float32 round-off of sums over up to 2e6 terms).  SURVEY appendix C
    asks 1e-3 WITHOUT freezing, which no float32 implementation -- torch-CPU included -- can meet (_fwd_bwd_check measures both)."""
    dev, _ = env
    es = _frozen_gradient_check(dev, kind, backbone, h, w, n, 0, 3e-4)
    assert es[len(es) // 2] < 1e-4, es[len(es) // 2]


@gpu
@pytest.mark.parametrize("split", [-1, 2])
def test_bf16_gradients_with_frozen_decisions_full_size(env, split, lbc_config):
    """(split: LBC_HDMAP_SPLIT -- -1 = the shipped policy, no split-K launch at this batch; 2 = every launch of the four-wave shape,
    layers 3 and 4, in two channel ranges: the split-K path with all three epilogue forms under the same bound.)
    The shipped bf16 mode (BASELINE.json config 3) at the reference's size, N = 32 = the per-GPU batch of the 8-GPU run, on a
    trained-like (warm-started) ResNet-34: every parameter gradient against the float64 oracle that takes the executor's own
    ReLU / max-pool decisions AND rounds where the executor rounds (MFMA operands, stored activations and activation gradients to
    bf16: oracle flags MFMA_BF16 / ACT_BF16) -- an ABSOLUTE statement about the mode, next to the autocast-relative one below.
    Bound: every tensor within BF16_FROZEN_MAX[its group] of its largest entry -- head, decoder, layers 2-4, stem + layer 1, each 1.5x
    the value measured for that group (round 4 had ONE bound at 2x the global maximum: loose enough to pass a 2x mis-scale in a layer-4
    BatchNorm bias) -- and the median tensor within BF16_FROZEN_MEDIAN (each stored tensor carries 2^-9 relative rounding noise and ~1e6
    such terms meet in one weight-gradient entry; the two evaluations round the same quantities but not bit-identical ones, so the
    noise does not cancel).  Inputs are seeded and the kernels deterministic: the measured values only move with the code."""
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS
    from learningbycheating_amd.training.native import NativeTrainer
    dev, _ = env
    n = 32
    lbc_config("LBC_HDMAP_SPLIT", split)
    rgb, speed, cmd = seeded_inputs("image", n, 71)
    onehot = O.one_hot(cmd)
    g = torch.Generator().manual_seed(73)
    tgt = torch.rand((n, 4, 5, 2), generator=g)
    tgt[..., 0] = tgt[..., 0] * 1.2 - 0.6
    tgt[..., 1] = tgt[..., 1] * 0.5 + 0.3
    torch.manual_seed(74)
    student = ImagePolicyModelSS("resnet34", all_branch=True).to(dev)
    warm = NativeTrainer(student, None, n, (3, 160, 384), dev, phase="l1_all", lr=1e-3)
    for _ in range(40):
        warm.step(rgb.to(dev), speed.to(dev), onehot.to(dev), target=tgt.to(dev))
    torch.cuda.synchronize()
    del warm
    sd = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
    es = _frozen_gradient_check(dev, "image", "resnet34", 160, 384, n, 2, BF16_FROZEN_MAX, sd=sd, seed=75)
    assert es[len(es) // 2] < BF16_FROZEN_MEDIAN, es[len(es) // 2]


#: bounds of test_bf16_gradients_with_frozen_decisions_full_size (rel-to-max per tensor): 1.5x the larger of the two measured arms
# (the shipped launch policy / LBC_HDMAP_SPLIT=2), profiles/r05_call5_bf16_accuracy_diag_and_fold_sweep.txt:
#   head 7.13e-2 / 6.87e-2, decoder 5.68e-2 / 4.96e-2, layers 2-4 5.35e-2 / 5.19e-2, stem + layer 1 3.21e-2 / 3.69e-2; median 2.30e-2 / 2.42e-2
BF16_FROZEN_MAX = {"head": 0.107, "decoder": 0.085, "layers2-4": 0.080, "stem+layer1": 0.055}
BF16_FROZEN_MEDIAN = 3.6e-2


@gpu
def test_bf16_gradients_match_autocast_reference(env):
    """Gradients of the shipped bf16 mode on a trained-like (warm-started) ResNet-34, decisions frozen, against the float64 oracle --
    next to the same statistic for the ORACLE run under torch's bf16 autocast (bf16 convolutions / activations, as the reference
    would run BASELINE.json config 3).  bf16 cannot meet an absolute 1e-3 / 2e-2 bar (each of ~100 stored tensors carries 2^-9
    relative rounding noise, and an untrained network amplifies it: 0.2 in the waypoints, profiles/r03_run4_frozen_grad_diag.txt);
    what it must meet is the accuracy of the reference in that precision: per-tensor errors no larger than 1.5x autocast's."""
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS
    from learningbycheating_amd.training.native import NativeTrainer
    dev, _ = env
    n, kind, backbone, h, w = 8, "image", "resnet34", 160, 384
    rgb, speed, cmd = seeded_inputs("image", n, 51)
    onehot = O.one_hot(cmd)
    g = torch.Generator().manual_seed(53)
    tgt = torch.rand((n, 4, 5, 2), generator=g)
    tgt[..., 0] = tgt[..., 0] * 1.2 - 0.6
    tgt[..., 1] = tgt[..., 1] * 0.5 + 0.3
    torch.manual_seed(54)
    student = ImagePolicyModelSS("resnet34", all_branch=True).to(dev)
    warm = NativeTrainer(student, None, n, (3, 160, 384), dev, phase="l1_all", lr=1e-3)
    for _ in range(40):
        warm.step(rgb.to(dev), speed.to(dev), onehot.to(dev), target=tgt.to(dev))
    torch.cuda.synchronize()
    del warm
    sd = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
    x, sp_, cm = seeded_inputs("image", n, 56)
    x = x.contiguous()               # (seeded_inputs hands out a permuted view of the uint8-style NHWC frames; the raw engine takes dense NCHW)
    oh = O.one_hot(cm)
    eng, tens = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=2)
    ps, pa = eng.forward(x.to(dev), sp_.to(dev), oh.to(dev), True)
    fz = frozen_decisions(eng)
    d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g), torch.randn((n, 5, 2), generator=g)
    eng.backward(d_sel.to(dev), d_all.to(dev))
    truth = O.as_params({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()})
    ts, ta = O.policy_forward(truth, kind, backbone, x.double(), sp_.double(), oh.double(), True, frozen=fz)
    ((ta * d_all.double()).sum() + (ts * d_sel.double()).sum()).backward()
    ac = O.as_params(sd)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        cs_, ca = O.policy_forward(ac, kind, backbone, x, sp_, oh, True, frozen=fz)
        ((ca.float() * d_all).sum() + (cs_.float() * d_sel).sum()).backward()
    e_hip, e_ac, c_hip, c_ac = [], [], [], []
    cos = lambda u, v: torch.nn.functional.cosine_similarity(u.reshape(1, -1).double(), v.reshape(1, -1).double()).item()
    for k, v in eng.grad_views.items():
        if k.startswith("location_pred") and k.endswith("bias"):
            continue
        ref = truth[k].grad
        e_hip.append(relerr(v.cpu().double(), ref)); e_ac.append(relerr(ac[k].grad.double(), ref))
        c_hip.append(cos(v.cpu(), ref)); c_ac.append(cos(ac[k].grad, ref))
    med = lambda z: sorted(z)[len(z) // 2]
    p90 = lambda z: sorted(z)[int(len(z) * 0.9)]
    f_hip = (pa.cpu().double() - ta.detach()).abs().max().item()
    f_ac = (ca.float().double() - ta.detach()).abs().max().item()
    _diag(dev, "bf16 executor vs float64 oracle (frozen decisions), warm-started r34 N=%d: waypoints %.2e, gradients rel-to-max median %.2e p90 %.2e max %.2e, "
               "cosine median %.4f min %.4f | oracle under torch bf16 autocast: waypoints %.2e, gradients median %.2e p90 %.2e max %.2e, cosine median %.4f min %.4f"
          % (n, f_hip, med(e_hip), p90(e_hip), max(e_hip), med(c_hip), min(c_hip), f_ac, med(e_ac), p90(e_ac), max(e_ac), med(c_ac), min(c_ac)))
    assert med(e_hip) <= 1.5 * med(e_ac) + 1e-3 and p90(e_hip) <= 1.5 * p90(e_ac) + 2e-3, (med(e_hip), med(e_ac), p90(e_hip), p90(e_ac))
    assert med(c_hip) >= med(c_ac) - 0.01 and min(c_hip) >= min(c_ac) - 0.05, (med(c_hip), med(c_ac), min(c_hip), min(c_ac))


@pytest.mark.parametrize("kind,backbone,h,w,n", [("birdview", "resnet18", 64, 64, 4), ("image", "resnet18", 32, 64, 5)])
def test_engine_small_emulated(env, kind, backbone, h, w, n):
    dev, _ = env
    # tiny spatial extents make BatchNorm ill-conditioned (a handful of samples per channel): loose gradient bound
    _fwd_bwd_check(dev, kind, backbone, h, w, n, 1e-4, 5e-3, flip_tol=2e-3)


@gpu
@pytest.mark.parametrize("kind,backbone,h,w,n", [("image", "resnet34", 160, 384, 4), ("birdview", "resnet18", 192, 192, 4),
                                                 ("image", "resnet18", 160, 384, 4)])
def test_engine_full_size(env, kind, backbone, h, w, n):
    dev, _ = env
    worst = _fwd_bwd_check(dev, kind, backbone, h, w, n, 1e-4, 1e-3, flip_tol=4e-2)
    print("worst relative gradient error", worst)


@gpu
@pytest.mark.parametrize("n,truth64", [(32, True), (64, False)])
def test_engine_full_size_at_baseline_batches(env, n, truth64):
    """BASELINE.json config 3's per-GPU batch (32 = 256 / 8) and config 2's batch (64) on the exact-f32 executor: forward in
    eval and training mode within 1e-4 of the oracle (north-star bar 1e-3), running statistics, every gradient.  At these
    batch sizes the tile policy picks the 128-row tiles the bench runs (at 4 images it picks 64 x 64)."""
    dev, _ = env
    worst = _fwd_bwd_check(dev, "image", "resnet34", 160, 384, n, 1e-4, 1e-3, flip_tol=4e-2, truth64=truth64)
    print("worst relative gradient error", worst)


@gpu
def test_forward_parity_at_bench_batch_256(env):
    """forward of the student (r34, 160x384) and the teacher (r18, 7x192x192) at the bench's batch of 256 images on the exact-f32
    executor vs the float32 oracle: |waypoints| error <= 1e-3 (north star), asserted at 2e-4"""
    dev, _ = env
    for kind, backbone, h, w, tol in (("image", "resnet34", 160, 384, 2e-4), ("birdview", "resnet18", 192, 192, 2e-4)):
        n = 256
        sd = O.make_state_dict(kind, backbone, 21, h, w)
        x, speed, cmd = _inputs(kind, n, h, w, 22)
        O.calibrate_running_stats(sd, kind, backbone, x[:32], speed[:32], cmd[:32])
        eng, tens = engine_from_state_dict(sd, kind, backbone, h, w, n, dev)
        for train in (False, True):
            ps, pa = eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), train)
            with torch.no_grad():
                os_, oa = O.policy_forward({k: v.clone() for k, v in sd.items()}, kind, backbone, x, speed, cmd, train)
            e = max((pa.cpu() - oa).abs().max().item(), (ps.cpu() - os_).abs().max().item())
            _diag(dev, "engine f32 %s %s N=256 train=%s: max |waypoint - oracle| = %.3e" % (kind, backbone, train, e))
            assert e < tol, (kind, train, e)
        del eng, tens
        torch.cuda.empty_cache()


@gpu
def test_modules_match_reference_fixtures(env):
    """ImagePolicyModelSS / BirdViewPolicyModelSS (the drop-in classes) reproduce what the real reference classes
    produced for the same seeded state_dict and inputs, within the 1e-3 bar of the north star (asserted at 1e-4)."""
    dev, _ = env
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS
    gold = torch.load(os.path.join(GOLD, "reference_outputs.pt"))
    for name, cls, kind, backbone in (("image_resnet34", ImagePolicyModelSS, "image", "resnet34"),
                                      ("birdview_resnet18", BirdViewPolicyModelSS, "birdview", "resnet18")):
        c = gold[name]
        sd = O.make_state_dict(kind, backbone, c["seed"])
        net = cls(backbone, all_branch=True)
        net.load_state_dict(sd, strict=True)
        net.to(dev)
        x, speed, cmd = seeded_inputs(kind, 2, c["input_seed"])
        onehot = O.one_hot(cmd)
        # the fixture IS a float32 evaluation; its own distance to a float64 evaluation bounds what can be asked
        sd64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        with torch.no_grad():
            _, o64 = O.policy_forward(sd64, kind, backbone, x.double(), speed.double(), onehot.double(), False)
        cond = (c["eval_preds"].double() - o64).abs().max().item()
        net.eval()
        with torch.no_grad():
            p, pa = net(x.to(dev), speed.to(dev), onehot.to(dev))
        e = (pa.cpu() - c["eval_preds"]).abs().max().item()
        assert e < max(1e-4, 5 * cond) and e < 1e-3, (name, e, cond)
        assert (p.cpu() - c["eval_pred"]).abs().max().item() < max(1e-4, 5 * cond)
        net.train()
        with torch.no_grad():
            p, pa = net(x.to(dev), speed.to(dev), onehot.to(dev))
        assert (pa.cpu() - c["train_preds"]).abs().max() < 1e-4
        got = net.state_dict()
        for k, v in c["running"].items():
            assert torch.allclose(got[k].cpu().float(), v.float(), rtol=1e-4, atol=1e-5), k
        net.all_branch = False
        net.eval()
        with torch.no_grad():
            assert net(x.to(dev), speed.to(dev), onehot.to(dev)).shape == (2, 5, 2)


@gpu
def test_phase1_step_gradients_vs_reference_fixture(env):
    """caller-style training step through autograd: loss.backward() reaches the HIP backward; gradients are compared
    with the values the real reference produced (sampled entries) and with the oracle (all entries)."""
    dev, _ = env
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS
    g = torch.load(os.path.join(GOLD, "reference_outputs.pt"))["phase1_step"]
    ssd = O.make_state_dict("image", "resnet34", g["student_seed"])
    tsd = O.make_state_dict("birdview", "resnet18", g["teacher_seed"])
    student = ImagePolicyModelSS("resnet34", all_branch=True)
    student.load_state_dict(ssd)
    teacher = BirdViewPolicyModelSS("resnet18", all_branch=True)
    teacher.load_state_dict(tsd)
    student.to(dev).train()
    teacher.to(dev).eval()
    rgb, speed, cmd = seeded_inputs("image", g["n"], g["rgb_seed"])
    bv, _, _ = seeded_inputs("birdview", g["n"], g["bv_seed"])
    onehot = O.one_hot(cmd).to(dev)
    with torch.no_grad():
        _, teac = teacher(bv.to(dev), speed.to(dev), onehot)
    assert (teac.cpu() - g["teacher_all"]).abs().max() < 1e-4
    _, pred_all = student(rgb.to(dev), speed.to(dev), onehot)
    assert (pred_all.detach().cpu() - g["pred_all"]).abs().max() < 1e-4
    # the reference's CoordConverter / LocationLoss arithmetic as torch ops on the device tensors (tiny)
    loss = O.phase1_loss(O.phase1_unproject(pred_all.cpu()), teac.cpu())
    assert torch.allclose(loss.detach(), g["loss"], rtol=2e-3)
    loss.mean().backward()
    named = dict(student.named_parameters())
    assert named["conv.fc.weight"].grad is None
    bad = 0
    for k, s in g["grads"].items():
        got = named[k].grad.detach().cpu().reshape(-1)[s["idx"]]
        if k.startswith("location_pred") and k.endswith("bias"):
            continue   # analytically ~0 (see test_engine_*), round-off only
        if not torch.allclose(got, s["val"], rtol=5e-2, atol=2e-2 * s["max"] + 1e-9):   # flip-tolerant, see _fwd_bwd_check
            bad += 1
    assert bad <= max(2, len(g["grads"]) // 10), "%d of %d gradient tensors deviate from the reference's values" % (bad, len(g["grads"]))


@pytest.mark.parametrize("size", ["small", pytest.param("full", marks=gpu)])
def test_loss_kernels(env, size):
    dev, _ = env
    from learningbycheating_amd.training.native import camera_struct
    n = 6 if size == "small" else 64
    g = torch.Generator().manual_seed(1)
    cam = torch.rand((n, 4, 5, 2), generator=g) * 1.6 - 0.8
    cam[..., 1] = cam[..., 1].abs() * 0.8 + 0.15
    teac = torch.rand((n, 4, 5, 2), generator=g) * 2 - 1
    lib = _lib.get()
    cs = camera_struct()
    # phase 1
    camr = cam.clone().requires_grad_(True)
    ref = O.phase1_loss(O.phase1_unproject(camr), teac)
    (ref.sum() * 0.25).backward()
    loss = torch.zeros(n, device=dev)
    d = torch.zeros_like(cam, device=dev)
    pc, tc = cam.to(dev), teac.to(dev)
    _lib.check(lib.lbc_loss(1, ctypes.byref(cs), _lib.ptr(pc), _lib.ptr(tc), n, 20, 0.25, _lib.ptr(loss), _lib.ptr(d), _lib.stream_for(pc)))
    assert torch.allclose(loss.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(d.cpu(), camr.grad, rtol=1e-4, atol=1e-7)
    # phase 0: student image-space prediction vs projected teacher map waypoints
    tmap = torch.rand((n, 5, 2), generator=g) * 2 - 1
    pred = (torch.rand((n, 5, 2), generator=g) * 2 - 1).requires_grad_(True)
    ref0 = O.phase0_loss(pred, O.phase0_project(tmap))
    (ref0.sum() * 0.5).backward()
    loss0 = torch.zeros(n, device=dev)
    d0 = torch.zeros((n, 5, 2), device=dev)
    pd, td = pred.detach().to(dev), tmap.to(dev)
    _lib.check(lib.lbc_loss(0, ctypes.byref(cs), _lib.ptr(pd), _lib.ptr(td), n, 5, 0.5, _lib.ptr(loss0), _lib.ptr(d0), _lib.stream_for(pd)))
    assert torch.allclose(loss0.cpu(), ref0.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(d0.cpu(), pred.grad, rtol=1e-5, atol=1e-8)
    # the same two kernels on the inputs the REAL reference classes were run on (tests/golden: phase0_loss, birdview_loss)
    gold = torch.load(os.path.join(GOLD, "reference_outputs.pt"))
    for kind, key, tkey, rows in ((0, "phase0_loss", "teacher_map", 5), (2, "birdview_loss", "gt", 5)):
        c = gold[key]
        m = c["pred"].shape[0]
        lg, dg = torch.zeros(m, device=dev), torch.zeros((m, 5, 2), device=dev)
        pg, tg = c["pred"].to(dev), c[tkey].to(dev)
        _lib.check(lib.lbc_loss(kind, ctypes.byref(cs), _lib.ptr(pg), _lib.ptr(tg), m, rows, 1.0 / m, _lib.ptr(lg), _lib.ptr(dg), _lib.stream_for(pg)))
        assert torch.allclose(lg.cpu(), c["loss"], rtol=1e-5, atol=1e-6), key
        assert torch.allclose(dg.cpu(), c["dpred"], rtol=1e-5, atol=1e-8), key
    # bird-view behaviour cloning L1 (pixel targets)
    gt = torch.rand((n, 5, 2), generator=g) * 192
    pred2 = (torch.rand((n, 5, 2), generator=g) * 2 - 1).requires_grad_(True)
    ref2 = O.birdview_loss(pred2, gt)
    ref2.sum().backward()
    loss2 = torch.zeros(n, device=dev)
    d2 = torch.zeros((n, 5, 2), device=dev)
    p2, g2 = pred2.detach().to(dev), gt.to(dev)
    _lib.check(lib.lbc_loss(2, ctypes.byref(cs), _lib.ptr(p2), _lib.ptr(g2), n, 5, 1.0, _lib.ptr(loss2), _lib.ptr(d2), _lib.stream_for(p2)))
    assert torch.allclose(loss2.cpu(), ref2.detach(), rtol=1e-5, atol=1e-6) and torch.allclose(d2.cpu(), pred2.grad, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("size", ["small", pytest.param("full", marks=gpu)])
def test_phase2_weight_kernel(env, size):
    """DAgger resampling weight (reference phase2_utils.py:50-59 as applied at train_image_phase2.py:203-206)"""
    dev, _ = env
    from learningbycheating_amd.training.native import camera_struct
    gold = torch.load(os.path.join(GOLD, "reference_outputs.pt"))["phase2_weight"]
    n = 6 if size == "small" else 128
    g = torch.Generator().manual_seed(3)
    cam = torch.rand((n, 5, 2), generator=g) * 1.6 - 0.8
    cam[..., 1] = cam[..., 1].abs() * 0.8 + 0.15
    teac = torch.rand((n, 5, 2), generator=g) * 2 - 1
    cam[:6], teac[:6] = gold["pred_cam"], gold["teacher"]
    cs = camera_struct()
    w = torch.zeros(n, device=dev)
    pc, tc = cam.to(dev), teac.to(dev)
    _lib.check(_lib.get().lbc_phase2_weight(ctypes.byref(cs), _lib.ptr(pc), _lib.ptr(tc), n, _lib.ptr(w), _lib.stream_for(pc)))
    assert torch.allclose(w.cpu(), O.phase2_weight(cam, teac), rtol=1e-5, atol=1e-7)
    assert torch.allclose(w.cpu()[:6], gold["weight"], rtol=1e-5, atol=1e-7)       # the real reference's get_weight


def test_replay_buffer_unnormalised_epoch_visits_every_sample_once():
    """reference train_image_phase2.py:170 DataLoader(shuffle=True, drop_last=True): before the weights are normalised an
    epoch is one permutation handed out in batch-size slices -- every sample's weight is written back exactly once"""
    from learningbycheating_amd.training.phase2_utils import ReplayBuffer
    buf = ReplayBuffer(torch.device("cpu"), buffer_limit=16, seed=2)
    buf.add_batch(torch.zeros((10, 2, 2, 3), dtype=torch.uint8), torch.zeros((10, 2, 2, 7), dtype=torch.uint8),
                  torch.ones(10), torch.zeros(10), [1.0] * 10)
    for epoch in range(2):
        buf.init_new_weights()
        seen = np.concatenate([buf.sample_indices(3) for _ in range(len(buf) // 3)])
        assert len(seen) == 9 and len(set(seen.tolist())) == 9 and set(seen.tolist()) <= set(range(10))


def test_replay_buffer_semantics():
    from learningbycheating_amd.training.phase2_utils import ReplayBuffer, repeat
    buf = ReplayBuffer(torch.device("cpu"), buffer_limit=6, seed=1)
    g = torch.Generator().manual_seed(0)
    buf.add_batch(torch.randint(0, 256, (8, 160, 384, 3), generator=g, dtype=torch.uint8), torch.zeros((8, 192, 192, 7), dtype=torch.uint8),
                  torch.tensor([1, 2, 3, 4, 1, 2, 3, 4]), torch.arange(8.0), [5, 1, 7, 3, 0.5, 9, 2, 8])
    assert len(buf) == 6 and sorted(buf._weights.tolist()) == [2, 3, 5, 7, 8, 9]         # lowest-loss samples evicted
    buf.init_new_weights()
    buf.update_weights([0, 5], torch.tensor([100.0, 200.0]))
    buf.normalize_weights()
    idx = buf.sample_indices(2000)
    assert set(idx.tolist()) <= set(range(6)) and (idx == 5).mean() > 0.5 and (idx == 0).mean() > 0.2   # loss-weighted resampling
    top, rgb, bv, cmd, speed = buf.get_highest_k(2)
    assert set(top.tolist()) == {0, 5} and rgb.shape == (2, 3, 160, 384) and rgb.max() <= 1.0
    assert torch.equal(repeat(torch.tensor([1, 2, 3]), 2), torch.tensor([1, 1, 2, 2, 3, 3]))


@pytest.mark.parametrize("size", ["small", pytest.param("full", marks=gpu)])
def test_fused_adam_matches_torch(env, size):
    dev, _ = env
    from learningbycheating_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(2)
    shapes = [(64, 3, 7, 7), (64,), (5, 64, 1, 1), (128, 64, 3, 3), (7,)] if size == "small" else [(512, 512, 3, 3), (640, 256, 3, 3), (64,), (1001,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ps = [p.contiguous(memory_format=torch.channels_last) if p.dim() == 4 else p for p in ps]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-3)
    mine = [(("p%d" % i), torch.nn.Parameter(p.clone().to(dev))) for i, p in enumerate(ps)]
    grads = {n: torch.zeros_like(p.data) for n, p in mine}
    fa = FusedAdam(mine, grads, lr=1e-3)
    for step in range(3):
        for (n, p), r in zip(mine, ref):
            gr = torch.randn(r.shape, generator=g)
            gr = gr.contiguous(memory_format=torch.channels_last) if gr.dim() == 4 else gr
            r.grad = gr.clone()
            grads[n].copy_(gr)
        opt.step()
        fa.step()
    for (n, p), r in zip(mine, ref):
        assert torch.allclose(p.data.cpu(), r.data, rtol=1e-5, atol=1e-6), n
        m, v = fa.state_of(n)
        st = opt.state[r]
        assert p.data.stride() == r.data.stride()
        assert torch.allclose(torch.as_strided(m.cpu(), r.shape, r.stride()), st["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(torch.as_strided(v.cpu(), r.shape, r.stride()), st["exp_avg_sq"], rtol=1e-5, atol=1e-9)


@gpu
def test_native_trainer_runs_and_is_deterministic(env):
    """the native phase-1 step (what bench.py times): finite loss, parameters move, bitwise repeatable"""
    dev, _ = env
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS
    from learningbycheating_amd.training.native import NativeTrainer
    ssd = O.make_state_dict("image", "resnet34", 31)
    tsd = O.make_state_dict("birdview", "resnet18", 32)
    rgb, speed, cmd = seeded_inputs("image", 4, 33)
    bv, _, _ = seeded_inputs("birdview", 4, 34)
    onehot = O.one_hot(cmd).to(dev)
    outs = []
    for rep in range(2):
        student = ImagePolicyModelSS("resnet34", all_branch=True)
        student.load_state_dict(ssd)
        teacher = BirdViewPolicyModelSS("resnet18", all_branch=True)
        teacher.load_state_dict(tsd)
        student.to(dev)
        teacher.to(dev)
        tr = NativeTrainer(student, teacher, 4, (3, 160, 384), dev, phase=1, lr=1e-4)
        losses = [tr.step(rgb.to(dev), speed.to(dev), onehot, birdview=bv.to(dev)).clone() for _ in range(3)]
        torch.cuda.synchronize()
        outs.append((torch.stack(losses).cpu(), student.conv.layer3[2].conv1.weight.detach().cpu().clone(), tr.eng.grad_flat.cpu().clone()))
    assert torch.isfinite(outs[0][0]).all()
    assert not torch.equal(outs[0][1], ssd["conv.layer3.2.conv1.weight"])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    # the fused loss kernel agrees with the oracle's loss arithmetic on the predictions of a fresh forward (the loss value
    # itself is ill-conditioned near the horizon pole, so it is not compared across two different forward evaluations)
    student = ImagePolicyModelSS("resnet34", all_branch=True)
    student.load_state_dict(ssd)
    teacher = BirdViewPolicyModelSS("resnet18", all_branch=True)
    teacher.load_state_dict(tsd)
    student.to(dev)
    teacher.to(dev)
    tr = NativeTrainer(student, teacher, 4, (3, 160, 384), dev, phase=1, lr=1e-4)
    loss = tr.step(rgb.to(dev), speed.to(dev), onehot, birdview=bv.to(dev), update=False).cpu()
    with torch.no_grad():
        _, teac = teacher.eval()(bv.to(dev), speed.to(dev), onehot)
    want = O.phase1_loss(O.phase1_unproject(tr.last_pred[1].cpu()), teac.cpu())
    assert torch.allclose(loss, want, rtol=1e-4, atol=1e-5), (loss, want)
    # phase-2 epoch on a tiny synthetic replay buffer: weights are written back, parameters move, nothing is NaN
    from learningbycheating_amd.bird_view.utils import bz_utils as bzu
    from learningbycheating_amd.training.train_image_phase2 import _train, synthetic_buffer
    import tempfile
    buf = synthetic_buffer(16, dev, seed=4)
    before = student.deconv[1].weight.detach().clone()
    with tempfile.TemporaryDirectory() as td:
        bzu.log.init(td)
        _train(buf, tr, {"device": dev, "batch_size": 4, "epoch_per_episode": 1, "speed_noise": 0.0, "log_iterations": 1,
                         "log_dir": td, "rank": 0}, episode=99)
    assert buf.normalized and (buf._weights != 1.0).any() and torch.isfinite(torch.as_tensor(buf._weights)).all()
    assert not torch.equal(before, student.deconv[1].weight.detach())


@pytest.mark.parametrize("glds", [False, True])
@pytest.mark.parametrize("kind,backbone,h,w,n", [("image", "resnet18", 64, 128, 4), pytest.param("image", "resnet34", 160, 384, 16, marks=gpu)])
def test_bn_backward_reduce_fused_into_dgrad_epilogue(env, kind, backbone, h, w, n, glds, lbc_config):
    """bf16 mode: the reduce pass of bn1's backward rides on conv2's input-gradient epilogue (conv_halo.hip for the 64-channel
    layer, conv_glds.hip when selected).  Same rounding points as the separate pass (sums of the stored bf16 gradient), only the
    order of the partial rows differs: every gradient tensor must agree with the unfused executor to f32 summation noise."""
    dev, _ = env
    if glds:
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
    sd = O.make_state_dict(kind, backbone, 11, h, w)
    x, speed, cmd = _inputs(kind, n, h, w, 9)
    g = torch.Generator().manual_seed(6)
    d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g), torch.randn((n, 5, 2), generator=g)
    grads, reduces = [], []
    for nofuse in (0, 1):
        lbc_config("LBC_NO_BN_BWD_FUSE", nofuse)
        eng, _ = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=2)
        eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), True)
        reduces.append(_launch_counts(lambda: eng.backward(d_sel.to(dev), d_all.to(dev))).get("bn_bwd_reduce", 0))
        grads.append({k: v.detach().cpu().clone() for k, v in eng.grad_views.items()})
    # Passes left: bn1's reduce rides on conv2's input gradient in every block; bn2's (round 5) on the input gradient of the NEXT block's
    # conv1 -- which IS the gradient wrt this block's output -- wherever that launch takes the persistent halo-staged kernel
    # (conv_hdmap_k<.., EPI 4>: mask with the block output, residual = the next block's identity gradient) and the next block has no
    # downsample.  Unfused: bn1 + bn2 per block, the three downsamples, and per decoder stage the BatchNorm and the ReLU / bias pass.
    nblk = {"resnet18": 8, "resnet34": 16}[backbone]
    assert reduces[1] == 2 * nblk + 9, reduces
    if glds:
        pairs = {"resnet18": 3, "resnet34": 10}[backbone]      # blocks of layers 2-4 behind a block without downsample
        assert reduces[0] == nblk + 9 - pairs, reduces
    elif h >= 160:
        # 16 images on the GPU: layer 2 (60 tiles of 256 x 128, 48-pixel rows) stays on the generic kernel, which fuses neither reduce:
        # bn1 of layers 1, 3, 4 (3 + 6 + 3) and bn2 behind the 5 + 2 plain blocks of layers 3 / 4 ride on input gradients
        assert reduces[0] == reduces[1] - (12 + 7), reduces
    else:
        assert reduces[0] < reduces[1], reduces
    rel = []
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        rel.append((a - b).abs().max().item() / (b.abs().max().item() + 1e-12))
    rel.sort()
    # a different order of f32 partial sums can flip a bf16 rounding of dY1 (relative step 2^-8) for isolated elements, which
    # the layers below amplify (measured on the 34-layer network at batch 16 with every eligible layer fused: worst tensor 2.2e-2
    # of its largest entry, median tensor 5.9e-3 = one to two bf16 steps; the 18-layer emulator case stays below 5e-3)
    assert rel[-1] < 5e-2 and rel[len(rel) // 2] < 1.5e-2, (rel[-1], rel[len(rel) // 2])


def _launch_counts(fn):
    """{kernel class: launches} of fn() under the library's launch profiler (lbc_profile_enable / lbc_profile_report)"""
    lib = _lib.get()
    lib.lbc_profile_enable(1)
    try:
        fn()
    finally:
        lib.lbc_profile_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    nbytes = lib.lbc_profile_report(buf, len(buf))
    return {ln.split()[0]: int(ln.split()[1]) for ln in buf.raw[:nbytes].decode().strip().splitlines()}


@pytest.mark.parametrize("kind,backbone,h,w,n,precision", [("image", "resnet18", 64, 128, 4, 2), ("birdview", "resnet18", 64, 64, 3, 0),
                                                           pytest.param("image", "resnet34", 160, 384, 32, 0, marks=gpu),
                                                           pytest.param("image", "resnet34", 160, 384, 32, 2, marks=gpu),
                                                           pytest.param("image", "resnet34", 160, 384, 16, 0, marks=gpu),
                                                           pytest.param("image", "resnet34", 160, 384, 16, 2, marks=gpu)])
def test_batchnorm_finalize_folded_into_its_consumer(env, kind, backbone, h, w, n, precision, lbc_config):
    """small per-GPU batches: where a BatchNorm's partial rows are few, the elementwise pass that consumes its coefficients does
    the finalize itself (BnApplyArgs::fold / BnBwdApplyArgs::fold; reference arithmetic resnet.py:38-54 forward and autograd).
    Against LBC_NO_BN_FOLD=1 (every finalize its own launch): fewer launches, and the same waypoints, running statistics, saved
    coefficients and gradients up to the order of the float64 row sums (<= 1e-6; in the bf16 mode a last-bit change of a
    coefficient can flip a bf16 rounding downstream: bounds of test_bn_backward_reduce_fused_into_dgrad_epilogue)."""
    dev, _ = env
    if precision == 2:
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
    sd = O.make_state_dict(kind, backbone, 23, h, w)
    x, speed, cmd = _inputs(kind, n, h, w, 24)
    g = torch.Generator().manual_seed(25)
    d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g), torch.randn((n, 5, 2), generator=g)
    runs = []
    for nofold in (1, 0):
        lbc_config("LBC_NO_BN_FOLD", nofold)
        eng, tens = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=precision)
        out = {}

        def step():
            out["pred"] = eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), True)
            eng.backward(d_sel.to(dev), d_all.to(dev))
        counts = _launch_counts(step)
        runs.append((counts, out["pred"][1].cpu().clone(), {k: v.detach().cpu().clone() for k, v in tens.items()},
                     {k: v.detach().cpu().clone() for k, v in eng.grad_views.items()},
                     {k: v.float().cpu().clone() for k, v in eng.activations().items() if k.endswith((".scale", ".shift"))}))
    (c0, p0, t0, g0, a0), (c1, p1, t1, g1, a1) = runs
    fin0 = c0.get("bn_finalize", 0) + c0.get("bn_bwd_finalize", 0)
    fin1 = c1.get("bn_finalize", 0) + c1.get("bn_bwd_finalize", 0)
    _diag(dev, "folded BatchNorm finalizes, precision %d %s %s %dx%d N=%d: finalize launches %d -> %d, all launches %d -> %d"
          % (precision, kind, backbone, h, w, n, fin0, fin1, sum(c0.values()), sum(c1.values())))
    assert fin1 <= fin0 - 20 and sum(c1.values()) <= sum(c0.values()) - 20, (c0, c1)
    tight = precision == 0
    # (f32: the coefficients agree to the last bit or two -- the order of the float64 row sums -- which at full size flips an
    #  isolated ReLU decision: waypoints to ~1e-6, the gradient tensors downstream of the flip to ~1e-3 of their largest entry)
    assert (p0 - p1).abs().max().item() < (5e-6 if tight else 2e-2)
    for k in a0:
        assert torch.allclose(a0[k], a1[k], rtol=1e-6 if tight else 2e-2, atol=1e-7 if tight else 1e-3), k
    for k in t0:
        if k.endswith(("running_mean", "running_var")):
            assert torch.allclose(t0[k], t1[k], rtol=1e-6 if tight else 1e-3, atol=1e-7 if tight else 1e-5), k
        if k.endswith("num_batches_tracked"):
            assert int(t0[k]) == int(t1[k]) == 1, k
    floor = 1e-6 * max(v.abs().max().item() for v in g0.values())        # (the head's biases: analytically zero gradients, round-off only)
    rels = sorted(((g0[k] - g1[k]).abs().max().item() / (g0[k].abs().max().item() + floor), k) for k in g0
                  if not (k.startswith("location_pred") and k.endswith("bias")))
    rel = [r for r, _ in rels]
    if tight and h < 160:
        assert rel[len(rel) // 2] < 2e-5 and rel[-1] < 5e-3, (rel[len(rel) // 2], rels[-3:])
    elif tight:
        # the reference-sized network has ~1e8 ReLU inputs per batch: a last-bit change of a scale / shift flips a handful of them, and
        # every flip moves the tensors upstream of it by ~1e-3 of their largest entry (measured: median 1.5e-4 .. 1.0e-3, max 1.5e-2 --
        # the flip statistics of _fwd_bwd_check, where two f32 evaluations of the SAME arithmetic differ by as much)
        assert rel[len(rel) // 2] < 5e-3 and rel[-1] < 5e-2, (rel[len(rel) // 2], rels[-3:])
    else:
        assert rel[-1] < 5e-2 and rel[len(rel) // 2] < 1.5e-2, (rel[-1], rel[len(rel) // 2])


@pytest.mark.parametrize("kind,backbone,h,w,n", [("image", "resnet18", 64, 128, 4), pytest.param("image", "resnet34", 160, 384, 32, marks=gpu),
                                                 pytest.param("image", "resnet34", 160, 384, 16, marks=gpu)])
def test_split_k_convolutions_inside_the_network(env, kind, backbone, h, w, n, lbc_config):
    """small per-GPU batches: the deep layers' 3x3 convolutions (resnet.py:38-54; 120 tiles for 256 CUs in layer 4 at 32 images) cut
    their channel contraction into ranges -- one workgroup per (tile, range), f32 partial tiles in the weight gradients' slab arena,
    a second launch that sums them and does the epilogue (statistics, residual, the fused BatchNorm-backward reduce; the folded
    BatchNorm of an eval forward).  Against LBC_HDMAP_SPLIT=0: the split launches exist, and waypoints / statistics / gradients agree
    within what two bf16 evaluations with regrouped f32 sums differ by (the kernel-level comparison -- within one bf16 rounding of the
    unsplit launch -- is test_conv_hdma_fwd_dgrad's).  Untrained networks are too ill-conditioned in bf16 for an A/B of their gradients:
    the small network's split arm is held against the float64 frozen-decision oracle here."""
    dev, _ = env
    small = h < 160
    if small:
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
        lbc_config("LBC_HDMA_CFG", 4)          # the four-wave shape for every eligible launch of the small network
    sd = O.make_state_dict(kind, backbone, 33, h, w)
    x, speed, cmd = _inputs(kind, n, h, w, 34)
    g = torch.Generator().manual_seed(35)
    d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g), torch.randn((n, 5, 2), generator=g)
    runs = []
    for split in (0, 2):                       # (2: every launch of the four-wave shape in two ranges -- layers 3 and 4 at the reference's size)
        lbc_config("LBC_HDMAP_SPLIT", split)
        eng, tens = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=2)
        out = {}

        def step():
            out["pred"] = eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), True)
            eng.backward(d_sel.to(dev), d_all.to(dev))
        counts = _launch_counts(step)
        grads = {k: v.detach().cpu().clone() for k, v in eng.grad_views.items()}
        stats = {k: v.detach().cpu().clone() for k, v in tens.items() if k.endswith(("running_mean", "running_var"))}
        # (eval forward: the folded BatchNorm + residual + ReLU epilogue.  With running statistics one step old the network does not
        #  normalize -- activations in the hundreds, a peaked soft-argmax: compared at the block outputs, relative to their scale)
        counts_eval = _launch_counts(lambda: eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), False))
        ev = {k: v.float().cpu().clone() for k, v in eng.activations().items() if k.startswith("conv.layer") and k.count(".") == 2}
        runs.append((counts, counts_eval, out["pred"][1].cpu().clone(), ev, grads, stats))
        if small and split:
            _frozen_gradient_check(dev, kind, backbone, h, w, n, 2, 0.35)      # (the bound of test_gradients_with_frozen_decisions_emulated)
    (c0, e0, p0, q0, g0, t0), (c1, e1, p1, q1, g1, t1) = runs
    qd = max(((q0[k] - q1[k]).abs().max() / q0[k].abs().max()).item() for k in q0)
    nsplit = [sum(v for k, v in c.items() if k.endswith("_split")) for c in (c0, e0, c1, e1)]
    floor = 1e-6 * max(v.abs().max().item() for v in g0.values())
    rel = sorted((g0[k] - g1[k]).abs().max().item() / (g0[k].abs().max().item() + floor) for k in g0
                 if not (k.startswith("location_pred") and k.endswith("bias")))
    _diag(dev, "split-K convolutions %s %s %dx%d N=%d: split launches per training step %d, per eval forward %d; |dwaypoint| train %.2e, eval block "
               "outputs (relative to their largest entry) %.2e; gradients split vs unsplit rel-to-max median %.2e max %.2e"
          % (kind, backbone, h, w, n, nsplit[2], nsplit[3], (p0 - p1).abs().max().item(), qd, rel[len(rel) // 2], rel[-1]))
    assert nsplit[0] == 0 and nsplit[1] == 0 and nsplit[2] >= 6 and nsplit[3] >= 3, (c0, c1, e1)
    assert sum(c0.values()) == sum(c1.values())              # (a bracket per convolution, split or not)
    assert (p0 - p1).abs().max().item() < (5e-2 if small else 0.15) and len(q0) >= 8 and qd < 3e-2
    for k in t0:           # (upstream bf16 roundings that fell the other way move a batch mean by ~1e-3 of what the step added to the initial 0 / 1)
        ref = (t0[k] - (1.0 if k.endswith("var") else 0.0)).abs().max().item()
        assert (t0[k] - t1[k]).abs().max().item() < 1e-2 * ref + 1e-6, k
    # (the gradients of an UNTRAINED network in bf16 are not comparable between two evaluations that round differently -- measured at the
    #  reference's size, 32 images: median 0.26 rel-to-max, the level of the small network here; the split path's gradients are held against
    #  the float64 frozen-decision oracle instead: above for the small network, test_bf16_gradients_with_frozen_decisions_full_size[2] on the GPU)


@pytest.mark.parametrize("precision", ["bf16"])
def test_frozen_teacher_derives_its_weight_copies_once(env, precision):
    """lbc_net_set_frozen (NativeTrainer sets it on the privileged teacher, train_image_phase1.py:244-248): the second eval-mode
    forward launches neither weight_prep nor bn_eval_prep and returns the same bits; load_state_dict derives everything again"""
    from learningbycheating_amd.bird_view.models import BirdViewPolicyModelSS
    dev, _ = env
    small = torch.device(dev).type != "cuda"
    hw = (64, 64) if small else (192, 192)
    torch.manual_seed(81)
    t = BirdViewPolicyModelSS("resnet18", all_branch=True, **({"input_hw": hw} if small else {}))
    t.precision = precision
    t = t.to(dev).eval()
    x, speed, cmd = _inputs("birdview", 3, hw[0], hw[1], 82)
    eng = t.engine((3, 7) + hw, dev, max_batch=3, with_grads=False)
    eng.set_frozen(True)
    outs, counts = [], []
    for _ in range(2):
        counts.append(_launch_counts(lambda: outs.append(eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), False)[1].cpu().clone())))
    assert counts[0].get("bn_eval_prep", 0) == 1 and counts[1].get("bn_eval_prep", 0) == 0, counts
    assert counts[1].get("weight_prep", 0) == 0 and (precision == "fp32" or counts[0].get("weight_prep", 0) == 1), counts
    assert torch.equal(outs[0], outs[1])
    # new weights through the module API: derived again
    sd = {k: (v * 1.5 if k.endswith("deconv.7.weight") else v.clone()) for k, v in t.state_dict().items()}
    t.load_state_dict(sd)
    c3 = _launch_counts(lambda: outs.append(eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), False)[1].cpu().clone()))
    assert c3.get("bn_eval_prep", 0) == 1 and not torch.equal(outs[2], outs[0])
    fresh = BirdViewPolicyModelSS("resnet18", all_branch=True, **({"input_hw": hw} if small else {}))
    fresh.precision = precision
    fresh.load_state_dict(sd)
    fresh = fresh.to(dev).eval()
    with torch.no_grad():
        want = fresh(x.to(dev), speed.to(dev), cmd.to(dev))[1].cpu()
    assert torch.equal(outs[2], want)


@pytest.mark.parametrize("precision", [0, 2])
@pytest.mark.parametrize("kind,backbone,h,w,n", [("image", "resnet18", 64, 128, 4), pytest.param("image", "resnet34", 160, 384, 16, marks=gpu)])
def test_staged_backward_equals_the_single_call(env, kind, backbone, h, w, n, precision, lbc_config):
    """the data-parallel trainer calls the backward stage by stage (head + decoder | layer 4 | ... | stem) and all-reduces a stage's
    gradient range behind each call: every stage's call must leave exactly what the one-call backward leaves -- in the bf16 mode a
    stage's weight gradients are deferred to the END of its call (grouped launches), in the f32 modes they ride a side stream"""
    dev, _ = env
    if precision == 2:
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
    sd = O.make_state_dict(kind, backbone, 19, h, w)
    x, speed, cmd = _inputs(kind, n, h, w, 13)
    g = torch.Generator().manual_seed(14)
    d_all = torch.randn((n, 4, 5, 2), generator=g)
    grads = []
    for staged in (False, True):
        eng, _ = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=precision)
        eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), True)
        if staged:
            for st in range(6):
                eng.backward(None, d_all.to(dev), st)
        else:
            eng.backward(None, d_all.to(dev))
        grads.append({k: v.detach().cpu().clone() for k, v in eng.grad_views.items()})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


@pytest.mark.parametrize("kind,backbone,h,w,nmax,n", [("image", "resnet18", 64, 128, 6, 2), pytest.param("image", "resnet34", 160, 384, 64, 24, marks=gpu)])
def test_batch_below_the_planned_maximum(env, kind, backbone, h, w, nmax, n, lbc_config):
    """bf16 mode: the workspace (dY arena, split-K slabs) is planned at max_batch, kernel choice and split counts follow the batch of the
    call -- a smaller batch on a larger plan must give what a plan of its own size gives (same kernels, same splits: bit-identical)"""
    dev, _ = env
    lbc_config("LBC_GEMM256_MIN_TILES", 1)
    sd = O.make_state_dict(kind, backbone, 17, h, w)
    x, speed, cmd = _inputs(kind, nmax, h, w, 12)
    g = torch.Generator().manual_seed(9)
    d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g), torch.randn((n, 5, 2), generator=g)
    grads = []
    for plan in (nmax, n):
        eng, _ = engine_from_state_dict(sd, kind, backbone, h, w, plan, dev, precision=2)
        if plan == nmax:      # the plan's own size first: leaves its traces in every buffer
            eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), True)
            eng.backward(torch.randn((nmax, 5, 2), generator=g).to(dev), torch.randn((nmax, 4, 5, 2), generator=g).to(dev))
        eng.forward(x[:n].contiguous().to(dev), speed[:n].to(dev), cmd[:n].to(dev), True)
        eng.backward(d_sel.to(dev), d_all.to(dev))
        grads.append({k: v.detach().cpu().clone() for k, v in eng.grad_views.items()})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


@pytest.mark.parametrize("precision", [1, 2, "2-tiles128", "2-glds"])
@pytest.mark.parametrize("kind,backbone,h,w,n", [("image", "resnet18", 64, 128, 4), pytest.param("image", "resnet34", 160, 384, 8, marks=gpu),
                                                 pytest.param("birdview", "resnet18", 192, 192, 8, marks=gpu)])
def test_engine_bf16_mfma_mode(env, kind, backbone, h, w, n, precision, lbc_config):
    """precision=1: convolution MFMA operands rounded to bf16, everything else f32.  Every kernel of this mode is checked
    tightly in tests/test_kernels.py against rounded-operand references; end to end the comparison can only be
    statistical, because a bf16 rounding boundary (relative step 2^-8) crossed by one element after a 1e-7 perturbation
    moves downstream activations by ~1e-2 (measured: the bf16-emulating oracle itself moves by 3e-2 under 1e-7 input
    noise on small shapes).  Checked: predictions close to the bf16-emulating oracle, gradients strongly aligned."""
    dev, _ = env
    if precision == "2-tiles128":
        # the tile policy of large batches (128-row tiles: the halo-staged layer-1 kernel, 128 x 64 / 128 x 128 igemm tiles)
        # on a test-sized batch
        lbc_config("LBC_FORCE_CFG", 0)
        precision = 2
    if precision == "2-glds":
        # the 8-wave LDS-DMA convolution (conv_glds.hip) for every stride-1 3x3 layer with >= 128 output channels, which
        # otherwise needs training-size batches to be selected
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
        precision = 2
    sd = O.make_state_dict(kind, backbone, 3, h, w)
    x, speed, cmd = _inputs(kind, n, h, w, 4)
    eng, tens = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=precision)
    ps, pa = eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), True)
    O.MFMA_BF16 = True
    O.ACT_BF16 = precision == 2   # precision 2: activations and their gradients are also stored as bf16
    try:
        sp = O.as_params(sd)
        ops, opa = O.policy_forward(sp, kind, backbone, x, speed, cmd, True)
        g = torch.Generator().manual_seed(5)
        d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g), torch.randn((n, 5, 2), generator=g)
        eng.backward(d_sel.to(dev), d_all.to(dev))
        ((opa * d_all).sum() + (ops * d_sel).sum()).backward()
    finally:
        O.MFMA_BF16 = O.ACT_BF16 = False
    xp = O.as_params(sd)                                                       # unrounded f32 forward / backward
    exs, exact = O.policy_forward(xp, kind, backbone, x, speed, cmd, True)
    ((exact * d_all).sum() + (exs * d_sel).sum()).backward()
    exact = exact.detach()
    err = (pa.cpu() - opa).abs().max().item()
    e_eng, e_emu = (pa.cpu() - exact).abs(), (opa.detach() - exact).abs()
    print("precision %d: |engine - f32| mean %.3e max %.3e; |emulation - f32| mean %.3e max %.3e; |engine - emulation| max %.3e"
          % (precision, e_eng.mean(), e_eng.max(), e_emu.mean(), e_emu.max(), err))
    # the executor's reduced-precision result must sit as close to the exact-f32 result as the emulation of its rounding
    # points does (two such evaluations differ from each other by as much as each differs from f32)
    assert e_eng.mean().item() < 2.0 * e_emu.mean().item() + 1e-3 and e_eng.max().item() < 3.0 * e_emu.max().item() + 1e-2
    # engine vs emulation: two evaluations with the same rounding points but different summation orders sit at most (their
    # distances to exact f32 added) apart.  This is an UNTRAINED network with seeded random BatchNorm affines, the worst case
    # for error growth; the accuracy the shipped mode is held to is asserted on a trained-like network in
    # test_bf16_mode_declared_accuracy (WAYPOINT_TOLERANCE['bf16'] = 1e-2).
    assert err < e_eng.max().item() + e_emu.max().item() + 1e-3, err
    assert (pa.cpu() - opa).abs().mean().item() < e_eng.mean().item() + e_emu.mean().item() + 1e-3
    def cosines(ga, gb):
        out = []
        for k in eng.grad_views:
            a, b = ga(k).reshape(-1).double(), gb(k).reshape(-1).double()
            if b.norm() > 1e-6 and not (k.startswith("location_pred") and k.endswith("bias")):
                out.append((torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item())
        out.sort()
        return out[len(out) // 2], out[len(out) // 10], out[0]

    g_eng, g_emu, g_f32 = (lambda k: eng.grad_views[k].cpu()), (lambda k: sp[k].grad), (lambda k: xp[k].grad)
    c_ee, c_ef, c_mf = cosines(g_eng, g_emu), cosines(g_eng, g_f32), cosines(g_emu, g_f32)
    print("precision %d gradient cosines (median, p10, min): engine~emulation %.3f %.3f %.3f | engine~f32 %.3f %.3f %.3f | emulation~f32 %.3f %.3f %.3f"
          % ((precision,) + c_ee + c_ef + c_mf))
    # Two reduced-precision evaluations of an untrained 34-layer BatchNorm network decorrelate (measured on MI355X, r34
    # 160x384 N=8: precision 1 median 0.88 / min 0.80; precision 2 median 0.74 / min 0.59); a wrong kernel gives ~0.  The
    # yardstick is the emulation itself: the executor's gradients must be as close to exact f32 as the emulation's are.
    assert c_ef[0] > c_mf[0] - 0.08 and c_ef[1] > c_mf[1] - 0.12, (c_ef, c_mf)
    assert c_ee[0] > (0.8 if precision == 1 else 0.6) and c_ee[1] > (0.6 if precision == 1 else 0.45), c_ee


@gpu
def test_bf16_mode_declared_accuracy(env):
    """The shipped mixed-precision mode (bench default, BASELINE.json config 3) on a trained-like network: the student is
    warm-started exactly as bench.py does it (L1 steps towards below-horizon targets, f32), then
      (1) eval- and training-mode waypoints of the bf16 executor stay within WAYPOINT_TOLERANCE['bf16'] of the f32 executor on the
          same weights, the f32 executor within 1e-4 of the f32 oracle, and the bf16 executor deviates from f32 no more than
          the oracle does under torch's own bf16 autocast (the reference run the way BASELINE.json config 3 would run it: bf16
          convolutions and activations, f32 BatchNorm statistics / softmax) -- i.e. the mode is as accurate as the reference in bf16;
      (2) training from that checkpoint follows the f32 loss curve as closely as an f32 run whose INPUT is perturbed by 1e-3
          does: 50 steps of the warm start's L1 objective (every step within 10 %) and 200 steps of the phase-1 objective."""
    import learningbycheating_amd as pkg
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS
    from learningbycheating_amd.training.native import NativeTrainer
    dev, _ = env
    n = 32
    rgb, speed, cmd = seeded_inputs("image", n, 41)
    bv, _, _ = seeded_inputs("birdview", n, 42)
    onehot = O.one_hot(cmd).to(dev)
    rgb, speed, bv = rgb.to(dev), speed.to(dev), bv.to(dev)
    g = torch.Generator().manual_seed(43)
    tgt = torch.rand((n, 4, 5, 2), generator=g)
    tgt[..., 0] = tgt[..., 0] * 1.2 - 0.6
    tgt[..., 1] = tgt[..., 1] * 0.5 + 0.3
    torch.manual_seed(44)
    student = ImagePolicyModelSS("resnet34", all_branch=True).to(dev)
    torch.manual_seed(45)
    teacher = BirdViewPolicyModelSS("resnet18", all_branch=True).to(dev)
    warm = NativeTrainer(student, None, n, (3, 160, 384), dev, phase="l1_all", lr=1e-3)
    for _ in range(40):
        warm.step(rgb, speed, onehot, target=tgt.to(dev))
    torch.cuda.synchronize()
    del warm
    ckpt = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}

    def fresh(precision):
        m = ImagePolicyModelSS("resnet34", all_branch=True)
        m.load_state_dict(ckpt)
        m.precision = precision
        return m.to(dev)

    # (1) forward accuracy on inputs the warm start has not seen: three batches
    tol = pkg.WAYPOINT_TOLERANCE["bf16"]
    for in_seed, train in [(sd_, tr_) for sd_ in (46, 146, 246) for tr_ in (False, True)]:
        x2, s2, c2 = seeded_inputs("image", n, in_seed)
        oh2 = O.one_hot(c2)
        outs = {}
        for prec in ("fp32", "bf16"):
            m = fresh(prec)
            m.train(train)
            with torch.no_grad():
                outs[(prec, train)] = m(x2.to(dev), s2.to(dev), oh2.to(dev))[1].cpu()
        with torch.no_grad():
            _, oa = O.policy_forward({k: v.clone() for k, v in ckpt.items()}, "image", "resnet34", x2, s2, oh2, train)
            with torch.autocast("cpu", dtype=torch.bfloat16):
                _, oc = O.policy_forward({k: v.clone() for k, v in ckpt.items()}, "image", "resnet34", x2, s2, oh2, train)
        e32 = (outs[("fp32", train)] - oa).abs().max().item()
        d = (outs[("bf16", train)] - outs[("fp32", train)]).abs()
        dc = (oc.float() - oa).abs()
        _diag(dev, "bf16 vs f32 executor, warm-started r34 N=%d inputs %d train=%s: |dwaypoint| max %.3e mean %.3e (declared %.0e); the oracle under torch "
                   "bf16 autocast vs its own f32: max %.3e mean %.3e (ratio of the maxima %.2f, of the means %.2f); f32 executor vs f32 oracle max %.2e"
              % (n, in_seed, train, d.max().item(), d.mean().item(), tol, dc.max().item(), dc.mean().item(), d.max().item() / dc.max().item(),
                 d.mean().item() / dc.mean().item(), e32))
        assert e32 < 1e-4, e32
        # Round 5: the bound is a RATIO to what the reference arithmetic itself loses under torch's bf16 autocast on the same weights and
        # the same batch, measured in this test (round 4 held the training-mode maximum to the constant 3e-2, which one seed met with 4 %
        # headroom -- 2.88e-2 -- while autocast sat at 3.98e-2 on that batch: the constant tested the batch, not the mode).  Maximum
        # within 1.25x of autocast's maximum (measured 0.6 - 0.9x), mean within 1.25x of autocast's mean (measured ~0.9x), on three
        # batches and in both modes; eval mode additionally within the declared absolute tolerance (measured 1.2e-2 of 3e-2), and the
        # mean within the declared mean tolerance in both
        assert d.max().item() <= 1.25 * dc.max().item() + 1e-3, ("bf16 waypoint maximum vs autocast", in_seed, train, d.max().item(), dc.max().item())
        assert d.mean().item() <= 1.25 * dc.mean().item() + 2e-4, ("bf16 waypoint mean vs autocast", in_seed, train, d.mean().item(), dc.mean().item())
        assert d.mean().item() <= pkg.WAYPOINT_MEAN_TOLERANCE["bf16"], ("bf16 waypoint mean", in_seed, train, d.mean().item())
        if not train:
            assert d.max().item() <= tol, ("bf16 eval-mode waypoint deviation", in_seed, d.max().item(), tol)
    # (2) loss curves from the common checkpoint, same data every step.
    # (a) The warm start's own objective (L1 towards below-horizon targets in camera space: well conditioned), 200 steps: the bf16 run
    #     stays within 10 % of the f32 run at EVERY step (measured: 1.4 % over the first 50).
    # (b) The phase-1 objective, 200 steps.  It unprojects with 1/y (train_image_phase1.py:43-64); on this synthetic teacher training
    #     drives far waypoints to the horizon (y -> 0.05) and around step 32 the trajectory bifurcates: ANY perturbation decides which
    #     way it goes.  The f32 executor fed rgb + 1e-3 * U(-1, 1) leaves the unperturbed f32 curve at step 32 (by up to 6x) -- exactly
    #     where the bf16 run leaves it (profiles/r03_run2_bf16_curves.*, r03_run5_*; scripts/diag_bf16_curve.py).  Past that point the runs are
    #     different realisations of a chaotic system: from one run to the next the bf16 arm ends at 0.02 or 0.09 and may or may not
    #     show a one-step spike (its waypoint noise of 1e-2 is 20 % of y = 0.05; torch's bf16 autocast of the reference has the same
    #     noise, part (1)), the f32 control ends at 0.03, f32 at 0.008.  Asserted: identical behaviour up to the bifurcation (every
    #     step within 10 % for the first 25, departure not earlier than the f32 control's), and a finite, descending curve after it.
    noise = (torch.rand(rgb.shape, generator=torch.Generator().manual_seed(47)) * 2 - 1).to(dev) * 1e-3
    steps_a, steps_b = 200, 200
    curves = {}
    for arm, prec, x in (("fp32", "fp32", rgb), ("fp32_eps", "fp32", (rgb + noise).clamp(0, 1)), ("bf16", "bf16", rgb)):
        ca = None
        if arm != "fp32_eps":
            m = fresh(prec)
            tr = NativeTrainer(m, None, n, (3, 160, 384), dev, phase="l1_all", lr=1e-4)
            ca = torch.stack([tr.step(rgb, speed, onehot, target=tgt.to(dev)).mean() for _ in range(steps_a)]).cpu()
            del tr
        m = fresh(prec)
        t = BirdViewPolicyModelSS("resnet18", all_branch=True)
        t.load_state_dict(teacher.state_dict())
        t.precision = prec
        t.to(dev)
        tr = NativeTrainer(m, t, n, (3, 160, 384), dev, phase=1, lr=1e-4)
        cb = torch.stack([tr.step(x, speed, onehot, birdview=bv).mean() for _ in range(steps_b)]).cpu()
        del tr
        curves[arm] = (ca, cb)
    a, b = curves["fp32"][0], curves["bf16"][0]
    rel = ((a - b).abs() / a.abs().clamp_min(1e-6)).max().item()
    _diag(dev, "warm-start L1 objective, %d steps from the warm start: f32 first/last %.4f/%.4f, bf16 %.4f/%.4f; max relative per-step difference %.3f"
          % (steps_a, a[0], a[-1], b[0], b[-1], rel))
    assert torch.isfinite(a).all() and torch.isfinite(b).all() and rel < 0.10, rel
    assert a[-5:].mean() < a[:5].mean() and b[-5:].mean() < b[:5].mean()
    f, c, h = curves["fp32"][1], curves["fp32_eps"][1], curves["bf16"][1]
    assert torch.isfinite(f).all() and torch.isfinite(c).all() and torch.isfinite(h).all()

    def departs(x):        # first step more than 10 % away from the f32 curve
        bad = ((x - f).abs() / f.abs().clamp_min(1e-6) > 0.10).nonzero()
        return int(bad[0]) if len(bad) else steps_b
    t_c, t_h = departs(c), departs(h)
    tail = lambda x: x[-50:].median().item()
    _diag(dev, "phase-1 objective, %d steps from the warm start: f32 %.4f -> %.4f (median of the last 50), f32 with 1e-3 input noise -> %.4f, bf16 -> %.4f; "
               "first step > 10 %% off the f32 curve: control %d, bf16 %d; largest step loss after step 30: f32 %.3f control %.3f bf16 %.3f"
          % (steps_b, f[0], tail(f), tail(c), tail(h), t_c, t_h, f[30:].max(), c[30:].max(), h[30:].max()))
    assert t_h >= 25 and t_h >= min(t_c, 25) - 3, ("bf16 leaves the f32 curve earlier than a 1e-3 input perturbation does", t_h, t_c)
    assert tail(h) < 0.5 * h[:5].mean().item() and h[-10:].mean() <= h[-60:-50].mean() * 1.5, ("bf16 run does not descend", tail(h), h[:5].mean().item())


@gpu
def test_bf16_phase1_fit_matches_f32_over_seeds(env):
    """Does the bf16 mode FIT as well as f32?  Round 3 compared one seed (bf16 ended at 2.0x the f32 loss, the f32 head kernels at
    0.96x) -- but the synthetic phase-1 objective is chaotic (1 / y unprojection, train_image_phase1.py:43-64, one fixed batch): over
    three seeds EVERY arm, the exact-f32 run with 1e-3 input noise included, lands between 0.2x and 20x of the clean f32 run
    (profiles/r04_run1_bf16_seeds.log), so a single ratio says nothing and "within 1.25x on every seed" holds for no arm at all.
    What can be asserted is distributional.  Twelve seeds (weights, data, teacher reseeded), 200 steps from an f32 warm start, three
    arms: f32, f32 + 1e-3 input noise (the control: what ANY rounding-sized perturbation does), bf16.  Every run must be finite and
    descend; the bf16 arm's median tail-loss ratio to f32 must not exceed 2.5x the control's (the median of 6 log-ratios with
    sigma ~ 1.1 has a standard error of e^0.55); and bf16 must beat-or-match f32 (<= 1.25x) on no fewer seeds than the control
    does, minus two."""
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS
    from learningbycheating_amd.training.native import NativeTrainer
    dev, _ = env
    n, steps, nseeds = 32, 200, 12
    tails = {"fp32": [], "fp32_eps": [], "bf16": []}
    firsts = []
    for seed in range(nseeds):
        base = 1000 * seed
        rgb, speed, cmd = seeded_inputs("image", n, base + 41)
        bv, _, _ = seeded_inputs("birdview", n, base + 42)
        onehot = O.one_hot(cmd).to(dev)
        rgb, speed, bv = rgb.to(dev), speed.to(dev), bv.to(dev)
        g = torch.Generator().manual_seed(base + 43)
        tgt = torch.rand((n, 4, 5, 2), generator=g)
        tgt[..., 0] = tgt[..., 0] * 1.2 - 0.6
        tgt[..., 1] = tgt[..., 1] * 0.5 + 0.3
        torch.manual_seed(base + 44)
        student = ImagePolicyModelSS("resnet34", all_branch=True).to(dev)
        torch.manual_seed(base + 45)
        teacher = BirdViewPolicyModelSS("resnet18", all_branch=True).to(dev)
        warm = NativeTrainer(student, None, n, (3, 160, 384), dev, phase="l1_all", lr=1e-3)
        for _ in range(40):
            warm.step(rgb, speed, onehot, target=tgt.to(dev))
        torch.cuda.synchronize()
        del warm
        ckpt = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
        noise = (torch.rand(rgb.shape, generator=torch.Generator().manual_seed(base + 47)) * 2 - 1).to(dev) * 1e-3
        for arm, prec, x in (("fp32", "fp32", rgb), ("fp32_eps", "fp32", (rgb + noise).clamp(0, 1)), ("bf16", "bf16", rgb)):
            m = ImagePolicyModelSS("resnet34", all_branch=True)
            m.load_state_dict(ckpt)
            m.precision = prec
            m = m.to(dev)
            t = BirdViewPolicyModelSS("resnet18", all_branch=True)
            t.load_state_dict(teacher.state_dict())
            t.precision = prec
            t.to(dev)
            tr = NativeTrainer(m, t, n, (3, 160, 384), dev, phase=1, lr=1e-4)
            curve = torch.stack([tr.step(x, speed, onehot, birdview=bv).mean() for _ in range(steps)]).cpu()
            del tr
            assert torch.isfinite(curve).all(), (seed, arm)
            tails[arm].append(curve[-20:].median().item())
            if arm == "fp32":
                firsts.append(curve[:3].mean().item())
    med = lambda z: sorted(z)[len(z) // 2]
    rb = [b / f for b, f in zip(tails["bf16"], tails["fp32"])]
    rc = [c / f for c, f in zip(tails["fp32_eps"], tails["fp32"])]
    _diag(dev, "phase-1 fit over %d seeds, tail loss (median of the last 20 of %d steps): f32 %s | f32 + 1e-3 input noise %s (ratio to f32: %s, median %.2f) | "
               "bf16 %s (ratio to f32: %s, median %.2f)"
          % (nseeds, steps, " ".join("%.4f" % v for v in tails["fp32"]), " ".join("%.4f" % v for v in tails["fp32_eps"]), " ".join("%.2f" % v for v in rc), med(rc),
             " ".join("%.4f" % v for v in tails["bf16"]), " ".join("%.2f" % v for v in rb), med(rb)))
    for arm in tails:
        assert all(t < 0.95 * f for t, f in zip(tails[arm], firsts)), (arm, tails[arm], firsts)       # every run descends
    assert med(rb) <= 2.5 * max(1.0, med(rc)), (med(rb), med(rc))
    assert sum(r <= 1.25 for r in rb) >= sum(r <= 1.25 for r in rc) - 2, (rb, rc)
    # round 5 (12 seeds instead of 6): the bf16 arm's WORST seed is no worse than twice the control's worst -- a mode that fits badly on
    # some inputs shows in its tail, not in its median (round 4: bf16 worst 1.19x, control worst 1.70x over six seeds)
    assert max(rb) <= 2.0 * max(1.0, max(rc)), ("bf16 worst tail-loss ratio vs the control's worst", max(rb), max(rc))


@pytest.mark.parametrize("kind,backbone,h,w,n", [("image", "resnet18", 64, 128, 3), pytest.param("image", "resnet34", 160, 384, 4, marks=gpu)])
def test_head_forward_on_mfma_matches_f32_head(env, kind, backbone, h, w, n, lbc_config):
    """bf16 activations: the waypoint head's projection runs on the bf16 MFMA with folded weights rounded to bf16; the
    LDS/f32 head kernel on the same bf16 decoder output and the same rounded weights must agree to f32 summation order"""
    dev, _ = env
    sd = O.make_state_dict(kind, backbone, 9, h, w)
    x, speed, cmd = _inputs(kind, n, h, w, 8)
    eng, _ = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=2)
    for train in (True, False):
        ps1, pa1 = eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), train)
        lbc_config("LBC_HEAD_NO_MFMA", 1)
        ps2, pa2 = eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), train)
        lbc_config("LBC_HEAD_NO_MFMA", -1)
        # f32 summation order over up to 3840 soft-argmax terms (measured 2.2e-5 at 40 x 96); a wrong projection gives > 1e-2
        assert (pa1 - pa2).abs().max().item() < 1e-4 and (ps1 - ps2).abs().max().item() < 1e-4


@pytest.mark.parametrize("precision", [0, 2])
@pytest.mark.parametrize("kind,backbone,h,w,n", [("image", "resnet18", 32, 64, 2), ("birdview", "resnet18", 32, 32, 2),
                                                 pytest.param("image", "resnet34", 160, 384, 4, marks=gpu)])
def test_uint8_nhwc_frames_equal_float_nchw_input(env, kind, backbone, h, w, n, precision):
    """lbc_net_forward_u8: the dataset's uint8 NHWC frames give exactly the reference path's float (x/255, NCHW) result"""
    dev, _ = env
    sd = O.make_state_dict(kind, backbone, 10, h, w)
    c = 3 if kind == "image" else 7
    g = torch.Generator().manual_seed(11)
    u8 = torch.randint(0, 256, (n, h, w, c), generator=g, dtype=torch.uint8)
    xf = (u8.float() / 255.0).permute(0, 3, 1, 2).contiguous()
    speed = torch.rand(n, generator=g) * 10
    cmd = torch.eye(4)[torch.randint(0, 4, (n,), generator=g)]
    eng, _ = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=precision)
    for train in ((False, True) if precision == 2 else (False,)):     # (the exact-f32 MFMA is slow under the CPU emulator)
        ps1, pa1 = eng.forward(xf.to(dev), speed.to(dev), cmd.to(dev), train)
        ps2, pa2 = eng.forward(u8.to(dev), speed.to(dev), cmd.to(dev), train)
        assert torch.equal(pa1, pa2) and torch.equal(ps1, ps2)


@gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_training_scripts_chain_phase0_to_phase1(env, tmp_path, precision):
    """the reference's workflow with its flag names: phase-0 script -> model-1.th -> phase-1 script --ckpt; config.json and
    .th files in the reference's layout (strict load into the oracle's key/shape table)"""
    import json
    from learningbycheating_amd.training import train_image_phase0, train_image_phase1
    d0, d1 = tmp_path / "p0", tmp_path / "p1"
    common = ["--synthetic", "16", "--batch_size", "4", "--iters_per_epoch", "3", "--max_epoch", "1", "--log_iterations", "1",
              "--precision", precision]
    # more, larger phase-0 steps than phase-1: phase 1's unprojection has a 1/y pole at the horizon, the reference always starts it
    # from a phase-0 checkpoint whose waypoints are below the horizon
    train_image_phase0.main(["--log_dir", str(d0), "--lr", "1e-3"] + [("12" if c == "3" else c) for c in common])
    ck = d0 / "model-1.th"
    assert ck.exists() and (d0 / "config.json").exists()
    train_image_phase1.main(["--log_dir", str(d1), "--ckpt", str(ck)] + common)
    cfg = json.loads((d1 / "config.json").read_text())
    assert cfg["model_args"]["backbone"] == "resnet34" and cfg["phase0_ckpt"] == str(ck)
    sd = torch.load(str(d1 / "model-1.th"), map_location="cpu")
    layout = O.state_dict_layout("image", "resnet34")
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, tuple(shape)) for k, shape in layout]
    assert all(torch.isfinite(v.float()).all() for v in sd.values())


@gpu
def test_baseline_config_1_phase0_256_frames_batch_8(env, tmp_path):
    """BASELINE.json configs[0] at its literal shape: train_image_phase0 (training/train_image_phase0.py:152-242), ImagePolicyModelSS,
    256 synthetic 160 x 384 frames, batch 8, 32 steps = one pass over the frames (the reference runs it on PyTorch-CPU as a plumbing
    check; this package has no CPU path by design, so the same loop runs on the GPU): config.json + model-1.th in the reference's
    layout, finite decreasing loss in the log"""
    import json
    from learningbycheating_amd.training import train_image_phase0
    d0 = tmp_path / "cfg1"
    train_image_phase0.main(["--log_dir", str(d0), "--synthetic", "256", "--batch_size", "8", "--iters_per_epoch", "32", "--max_epoch", "1",
                             "--log_iterations", "8", "--lr", "1e-3"])
    cfg = json.loads((d0 / "config.json").read_text())
    assert cfg["model_args"]["backbone"] == "resnet34" and cfg["data_args"]["batch_size"] == 8
    sd = torch.load(str(d0 / "model-1.th"), map_location="cpu")
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, tuple(shape)) for k, shape in O.state_dict_layout("image", "resnet34")]
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    log = [json.loads(l) for l in open(d0 / "log.jsonl")]
    # (epoch 0 is the reference's no-update pass, train_image_phase0.py:165,202: the first record is the untrained loss)
    assert all(np.isfinite(r["train_loss_mean"]["mean"]) for r in log) and log[-1]["train_loss_mean"]["mean"] < log[0]["train_loss_mean"]["mean"]


@gpu
@pytest.mark.parametrize("precision", [0, 2])
def test_side_stream_backward_is_bit_identical_to_single_stream(env, lbc_config, precision):
    """the residual blocks' weight gradients run on an internal side stream; every kernel is deterministic, so a missing
    dependency (a buffer rewritten while a weight gradient still reads it) shows up as a bit difference against the
    single-stream order (LBC_NO_SIDE_STREAM=1), repeated a few times"""
    dev, _ = env
    kind, backbone, h, w, n = "image", "resnet34", 160, 384, 8
    sd = O.make_state_dict(kind, backbone, 12, h, w)
    x, speed, cmd = _inputs(kind, n, h, w, 13)
    g = torch.Generator().manual_seed(14)
    d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g).to(dev), torch.randn((n, 5, 2), generator=g).to(dev)

    def grads(single_stream):
        lbc_config("LBC_NO_SIDE_STREAM", 1 if single_stream else -1)     # read when the network is created
        eng, _ = engine_from_state_dict(sd, kind, backbone, h, w, n, dev, precision=precision)
        out = []
        for _ in range(3):
            eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), True)
            for st in range(eng.num_stages()):
                eng.backward(d_sel, d_all, st)
            torch.cuda.synchronize()
            out.append(eng.grad_flat.clone())
        return out

    ref = grads(True)
    got = grads(False)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    assert torch.equal(ref[0], ref[1]) and torch.equal(got[0], got[2])


# ---- misuse the reference's nn.Module API tolerates: must be an error here, never silently wrong numbers -------------------
def test_backward_through_a_stale_forward_is_an_error(env):
    """the executor keeps ONE workspace: p1 = net(x1); p2 = net(x2); p1.sum().backward() would differentiate the second
    forward's activations -- it raises instead (the reference's loops always backward before the next forward)"""
    dev, _ = env
    from learningbycheating_amd.bird_view.models import BirdViewPolicyModelSS
    torch.manual_seed(0)
    net = BirdViewPolicyModelSS("resnet18", all_branch=True)
    for b in range(4):                                    # a 32 x 32 map (8 x 8 soft-argmax grid) keeps the emulated run short;
        px, py = O.softmax_positions(8, 8)                # the public forward() insists on the reference's 192 x 192
        net.location_pred[b][2].pos_x, net.location_pred[b][2].pos_y = px, py
    net.precision = "bf16"                                # (the exact-f32 MFMA is the slowest thing to emulate)
    net = net.to(dev).train()
    x1, s1, c1 = _inputs("birdview", 2, 32, 32, 1)
    x2, s2, c2 = _inputs("birdview", 2, 32, 32, 2)
    p1, _ = net._run(x1.to(dev), s1.to(dev), c1.to(dev))
    p1.sum().backward()                                   # fine: no forward in between
    g1 = net.conv.conv1.weight.grad.clone()
    net.zero_grad()
    p1, _ = net._run(x1.to(dev), s1.to(dev), c1.to(dev))
    p2, _ = net._run(x2.to(dev), s2.to(dev), c2.to(dev))
    with pytest.raises(RuntimeError, match="stale forward"):
        p1.sum().backward()
    p2.sum().backward()                                   # the latest forward is still differentiable
    assert torch.isfinite(net.conv.conv1.weight.grad).all() and g1.abs().max() > 0


def test_engine_forward_validates_what_the_c_abi_will_dereference(env):
    dev, _ = env
    sd = O.make_state_dict("image", "resnet18", 3, 32, 64)
    eng, _ = engine_from_state_dict(sd, "image", "resnet18", 32, 64, 2, dev)
    x, speed, cmd = _inputs("image", 2, 32, 64, 4)
    x, speed, cmd = x.to(dev), speed.to(dev), cmd.to(dev)
    eng.forward(x, speed, cmd, False)
    bad = [
        (x.double(), speed, cmd, "float32"),                                                   # dtype
        (x.contiguous(memory_format=torch.channels_last), speed, cmd, "contiguous"),           # layout
        (x[:, :, :16].contiguous(), speed, cmd, "shape"),                                      # extent != plan
        (torch.cat([x, x, x]), torch.cat([speed] * 3), torch.cat([cmd] * 3), "batch"),         # > max_batch
        (x, speed.double(), cmd, "velocity"),
        (x, speed, cmd[:, :3].contiguous(), "command"),
        ((x * 255).to(torch.uint8), speed, cmd, "uint8"),                                      # uint8 frames must be NHWC
    ]
    for xi, si, ci, what in bad:
        with pytest.raises(RuntimeError):
            eng.forward(xi, si, ci, False)
    eng.forward(x, speed, cmd, True)
    with pytest.raises(RuntimeError, match="d_all"):
        eng.backward(None, torch.zeros((3, 4, 5, 2), device=dev))                              # batch of the last forward is 2


def test_flat_gradient_and_adam_state_are_16_byte_aligned(env):
    """every tensor's slice of the flat gradient buffer / Adam moments starts on a 256-byte boundary (adam_k and the
    all-reduce buckets use 16-byte accesses; the 5-element head biases used to misalign everything behind them)"""
    dev, _ = env
    from learningbycheating_amd.optim import FusedAdam
    from learningbycheating_amd.bird_view.models import BirdViewPolicyModelSS
    from learningbycheating_amd.parallel import stage_ranges
    net = BirdViewPolicyModelSS("resnet18", all_branch=True).to(dev)
    eng = net.engine((1, 7, 192, 192), dev, max_batch=1, with_grads=True)
    assert all(off % 64 == 0 for off, _ in eng.grad_offsets.values())
    assert all(v.data_ptr() % 16 == 0 for v in eng.grad_views.values())
    opt = FusedAdam(list(net.named_parameters()), eng.grad_views)
    assert all(off % 64 == 0 for off, _ in opt.offsets.values())
    r = sorted(stage_ranges(eng.grad_spans))
    assert r[0][0] == 0 and r[-1][1] == eng.grad_flat.numel() and all(b == c for (_, b), (c, _) in zip(r[:-1], r[1:]))
    # the pads are never written: a step on zero gradients leaves them (and the moments there) at zero
    opt.step()
    used = torch.zeros(eng.grad_flat.numel(), dtype=torch.bool)
    for off, n in eng.grad_offsets.values():
        used[off:off + n] = True
    assert opt.exp_avg.cpu()[~used].abs().max().item() == 0.0


@gpu
def test_training_scripts_birdview_phase2_and_lmdb_dataset(env, tmp_path):
    """the secondary scripts end to end with the reference's flag names: train_birdview (BASELINE config 4), phase 1 fed from
    an LMDB dataset in the reference's on-disk format with GPU augmentation and --batch_aug, train_image_phase2 (config 5);
    .th files in the reference's state_dict layout, validation pass logged"""
    import json
    from learningbycheating_amd.bird_view.utils.datasets.image_lmdb import write_synthetic_dataset
    from learningbycheating_amd.training import train_birdview, train_image_phase0, train_image_phase1, train_image_phase2
    data = write_synthetic_dataset(str(tmp_path / "data"), episodes=2, frames=48, seed=1)
    common = ["--batch_size", "4", "--iters_per_epoch", "3", "--max_epoch", "1", "--log_iterations", "1"]
    db = tmp_path / "bv"
    # the reference's default flags (5 px / 5 degrees of jitter: rotation + window on the GPU), then with the frame cap
    train_birdview.main(["--log_dir", str(db), "--dataset_dir", data] + common)
    train_birdview.main(["--log_dir", str(db), "--dataset_dir", data, "--x_jitter", "0", "--y_jitter", "3", "--angle_jitter", "0", "--max_frames", "20"] + common)
    sd = torch.load(str(db / "model-1.th"), map_location="cpu")
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, tuple(s)) for k, s in O.state_dict_layout("birdview", "resnet18")]
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    log = [json.loads(l) for l in open(db / "log.jsonl")]
    assert all("val_loss_mean" in r and "train_loss_mean" in r for r in log), "every epoch logs a training and a validation pass"
    # phase 0 on the LMDB dataset (teacher = the checkpoint just written), then phase 1 with augmentation + batch_aug
    d0, d1 = tmp_path / "p0", tmp_path / "p1"
    train_image_phase0.main(["--log_dir", str(d0), "--dataset_dir", data, "--teacher_path", str(db / "model-1.th"), "--lr", "1e-3",
                             "--augment", "super_hard"] + [("12" if c == "3" else c) for c in common])
    train_image_phase1.main(["--log_dir", str(d1), "--dataset_dir", data, "--ckpt", str(d0 / "model-1.th"), "--teacher_path", str(db / "model-1.th"),
                             "--augment", "super_hard", "--batch_aug", "2"] + common)
    sd1 = torch.load(str(d1 / "model-1.th"), map_location="cpu")
    assert [(k, tuple(v.shape)) for k, v in sd1.items()] == [(k, tuple(s)) for k, s in O.state_dict_layout("image", "resnet34")]
    assert all(torch.isfinite(v.float()).all() for v in sd1.values())
    cfg = json.loads((d1 / "config.json").read_text())
    assert cfg["data_args"]["batch_aug"] == 2 and cfg["data_args"]["dataset_dir"] == data
    # phase 2: two short episodes on a synthetic replay buffer, model-1.th saved (SAVE_EPISODES)
    d2 = tmp_path / "p2"
    train_image_phase2.main(["--log_dir", str(d2), "--ckpt", str(d1 / "model-1.th"), "--teacher_path", str(db / "model-1.th"), "--batch_size", "4",
                             "--synthetic", "16", "--max_episode", "2", "--epoch_per_episode", "1", "--log_iterations", "1"])
    assert (d2 / "config.json").exists()
    saved = sorted(p.name for p in d2.glob("model-*.th"))
    assert saved, "phase 2 saved no checkpoint"
    sd2 = torch.load(str(d2 / saved[0]), map_location="cpu")
    assert list(sd2.keys()) == list(sd1.keys())


@pytest.mark.parametrize("kind,backbone,h,w", [pytest.param("image", "resnet34", 160, 384, marks=gpu), pytest.param("birdview", "resnet18", 192, 192, marks=gpu)])
def test_batch1_inference_session_matches_reference_forward(env, kind, backbone, h, w):
    """the agent-side path (reference image.py:124-139): uint8 frame in, (5,2) waypoints out, batch 1, eval mode; equal to the
    oracle's forward on ToTensor(frame) within 1e-4 (north star 1e-3), and the hipGraph replay is bit-identical to eager launches"""
    import time
    from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS
    from learningbycheating_amd.inference import PolicySession
    dev, _ = env
    sd = O.make_state_dict(kind, backbone, 51)
    x, speed, cmd = _inputs(kind, 8, h, w, 52)
    O.calibrate_running_stats(sd, kind, backbone, x, speed, cmd)
    net = (ImagePolicyModelSS if kind == "image" else BirdViewPolicyModelSS)(backbone, all_branch=True)
    net.load_state_dict(sd)
    ses_g, ses_e = PolicySession(net, dev, use_graph=True), None
    net2 = (ImagePolicyModelSS if kind == "image" else BirdViewPolicyModelSS)(backbone, all_branch=True)
    net2.load_state_dict(sd)
    ses_e = PolicySession(net2, dev, use_graph=False)
    g = torch.Generator().manual_seed(53)
    c = 3 if kind == "image" else 7
    worst = 0.0
    for step in range(4):
        frame = torch.randint(0, 256, (h, w, c), generator=g, dtype=torch.uint8)
        if kind == "birdview":
            frame = (frame > 230).to(torch.uint8) * 255
        v, k = float(torch.rand(1, generator=g) * 10), int(torch.randint(1, 5, (1,), generator=g))
        got_g, got_e = ses_g.run_step(frame.numpy(), v, k), ses_e.run_step(frame.numpy(), v, k)
        assert np.array_equal(got_g, got_e), "graph replay differs from eager launches"
        xin = (frame.float() / 255.0).permute(2, 0, 1)[None]
        with torch.no_grad():
            want, _ = O.policy_forward({kk: vv.clone() for kk, vv in sd.items()}, kind, backbone, xin, torch.tensor([v]), O.one_hot(torch.tensor([float(k)])), False)
        worst = max(worst, float(np.abs(got_g - want[0].numpy()).max()))
    assert worst < 1e-4, worst
    lat = {}
    for name, ses in (("graph", ses_g), ("eager", ses_e)):
        frame = np.zeros((h, w, c), np.uint8)
        for _ in range(5):
            ses.run_step(frame, 3.0, 2)
        t0 = time.perf_counter()
        for _ in range(50):
            ses.run_step(frame, 3.0, 2)
        lat[name] = (time.perf_counter() - t0) / 50 * 1e3
    # the same session in the bf16 mode (what a deployed agent would run): within the declared tolerance of the f32 session
    from learningbycheating_amd import WAYPOINT_TOLERANCE
    net3 = (ImagePolicyModelSS if kind == "image" else BirdViewPolicyModelSS)(backbone, all_branch=True)
    net3.load_state_dict(sd)
    net3.precision = "bf16"
    ses_b = PolicySession(net3, dev, use_graph=True)
    frame = torch.randint(0, 256, (h, w, c), generator=g, dtype=torch.uint8)
    if kind == "birdview":
        frame = (frame > 230).to(torch.uint8) * 255
    db = float(np.abs(ses_b.run_step(frame.numpy(), 4.0, 3) - ses_g.run_step(frame.numpy(), 4.0, 3)).max())
    assert db < 3 * WAYPOINT_TOLERANCE["bf16"], db     # (an uncalibrated random network: looser than the trained-like bound)
    zf = np.zeros((h, w, c), np.uint8)
    for _ in range(5):
        ses_b.run_step(zf, 3.0, 2)
    t0 = time.perf_counter()
    for _ in range(50):
        ses_b.run_step(zf, 3.0, 2)
    lat["bf16"] = (time.perf_counter() - t0) / 50 * 1e3
    _diag(dev, "batch-1 inference %s %s: f32 |waypoint - oracle| max %.2e, bf16 vs f32 %.2e; latency per run_step (H2D + forward + D2H): "
               "f32 hipGraph %.3f ms, f32 eager %.3f ms, bf16 hipGraph %.3f ms"
          % (kind, backbone, worst, db, lat["graph"], lat["eager"], lat["bf16"]))
