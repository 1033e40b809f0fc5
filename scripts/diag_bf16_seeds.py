"""GPU diagnostic: is the bf16 mode's phase-1 fit as good as the f32 fit?  Several seeds, several arms.

    python scripts/diag_bf16_seeds.py [seeds] [steps] [batch] [out.json]

Round 3's single-seed run ended at 2.0x the f32 loss with the MFMA head and at 0.96x with the f32 head kernels -- but the same f32 run
with 1e-3 input noise ended at 3.2x: one seed of a chaotic objective (the 1/y unprojection of training/train_image_phase1.py:43-64)
says nothing.  Per seed (weights, data, teacher all reseeded), from a common f32 warm start, `steps` phase-1 steps per arm:
    fp32            exact-f32 executor
    fp32_eps        the same, rgb perturbed by 1e-3 * U(-1, 1) (the yardstick: what rounding-sized noise does to an f32 run)
    bf16            the shipped mode (MFMA head, folded head weights as a bf16 high + low pair)
    bf16_nosplit    round 3's head (one bf16 copy of the folded weights, LBC_HEAD_NO_SPLIT=1)
    bf16_f32head    the f32 head kernels on the bf16 decoder output (LBC_HEAD_NO_MFMA=1)
    bf16_wire       bf16 + every gradient bucket rounded to bf16 before Adam (what a bf16 all-reduce does on one rank)
Reported per arm: median over the last 20 steps of the mean loss, per seed, and the ratio to the f32 arm of the same seed."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lbc_oracle as O                      # noqa: E402  (diagnostic script: test infrastructure may be imported here)
from oracle.make_golden import seeded_inputs           # noqa: E402
from learningbycheating_amd import _lib                # noqa: E402
from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS   # noqa: E402
from learningbycheating_amd.training.native import NativeTrainer                                  # noqa: E402

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
out = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/bf16_seeds.json"
dev = torch.device("cuda", 0)

ARMS = [("fp32", "fp32", None, False, False), ("fp32_eps", "fp32", None, True, False), ("bf16", "bf16", None, False, False),
        ("bf16_nosplit", "bf16", ("LBC_HEAD_NO_SPLIT", 1), False, False), ("bf16_f32head", "bf16", ("LBC_HEAD_NO_MFMA", 1), False, False),
        ("bf16_wire", "bf16", None, False, True)]


class _WireRounding:
    """stands in for StageAllReducer on one rank: the bucket of a stage is rounded to bf16 in place (cast, identity all-reduce, cast back)"""

    def __init__(self, real):
        self.real = real

    def launch(self, stage):
        lo, hi = self.real.ranges[stage]
        self.real.flat[lo:hi] = self.real.flat[lo:hi].bfloat16().float()

    def fence(self):
        pass

    def wait(self):
        pass


res = {}
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
    for name, prec, opt, eps, wire in ARMS:
        if opt:
            _lib.config_set(opt[0], opt[1])
        m = ImagePolicyModelSS("resnet34", all_branch=True)
        m.load_state_dict(ckpt)
        m.precision = prec
        m = m.to(dev)
        t = BirdViewPolicyModelSS("resnet18", all_branch=True)
        t.load_state_dict(teacher.state_dict())
        t.precision = prec
        t.to(dev)
        tr = NativeTrainer(m, t, n, (3, 160, 384), dev, phase=1, lr=1e-4)
        if wire:
            tr.reducer = _WireRounding(tr.reducer)
        x = (rgb + noise).clamp(0, 1) if eps else rgb
        mean, ymin = [], []
        for _ in range(steps):
            l = tr.step(x, speed, onehot, birdview=bv)
            mean.append(l.mean().item())
            ymin.append(tr.last_pred[1][..., 1].min().item())
        res.setdefault(name, []).append(dict(mean=mean, ymin=ymin))
        del tr
        if opt:
            _lib.config_set(opt[0], -1)
        tail = sorted(mean[-20:])[10]
        print("seed %d %-13s first %.4f tail(median of last 20) %.4f max %.4f at step %d; min y %.4f"
              % (seed, name, mean[0], tail, max(mean), mean.index(max(mean)), min(ymin)), flush=True)
tails = {name: [sorted(r["mean"][-20:])[10] for r in runs] for name, runs in res.items()}
print("\ntail loss (median of the last 20 steps), per seed, and ratio to the f32 arm of the same seed:")
for name, v in tails.items():
    ratio = [a / b for a, b in zip(v, tails["fp32"])]
    print("%-13s %s   ratio %s   geometric mean ratio %.3f" % (name, " ".join("%.4f" % a for a in v), " ".join("%.2f" % r for r in ratio),
                                                            float(torch.tensor(ratio).log().mean().exp())))
json.dump(res, open(out, "w"))
