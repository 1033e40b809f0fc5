"""GPU diagnostic: phase-1 loss curves from a common warm start, per precision mode and with controls.

    python scripts/diag_bf16_curve.py [steps] [batch] [out.json]

Arms (same checkpoint, same batch every step -- the set-up of tests/test_model.py::test_bf16_mode_declared_accuracy):
    fp32            exact-f32 executor (the parity path)
    fp32_eps        the same with the rgb input perturbed by 1e-3 * U(-1, 1): how far two f32 runs that differ by rounding-sized
                    noise drift apart under this objective (the yardstick for "bf16 follows the f32 curve")
    bf16_mfma       bf16 MFMA operands, f32 activation storage
    bf16            the shipped mode (bf16 operands + bf16 activation storage)
    bf16_f32head    bf16 with the f32 waypoint-head kernels (LBC_HEAD_NO_MFMA=1)
Per step: mean loss, max per-sample loss, the smallest predicted camera-space y (the 1/y unprojection of
training/train_image_phase1.py:43-64 amplifies noise as y -> 0).
"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/bf16_curves.json"
dev = torch.device("cuda", 0)
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
noise = (torch.rand(rgb.shape, generator=torch.Generator().manual_seed(47)) * 2 - 1).to(dev) * 1e-3

ARMS = [("fp32", "fp32", None, False), ("fp32_eps", "fp32", None, True), ("bf16_mfma", "bf16_mfma", None, False),
        ("bf16", "bf16", None, False), ("bf16_f32head", "bf16", ("LBC_HEAD_NO_MFMA", 1), False)]
res = {}
for name, prec, opt, eps in ARMS:
    if opt:
        _lib.check(_lib.get().lbc_config_set(opt[0].encode(), opt[1]))
    m = ImagePolicyModelSS("resnet34", all_branch=True)
    m.load_state_dict(ckpt)
    m.precision = prec
    m = m.to(dev)
    t = BirdViewPolicyModelSS("resnet18", all_branch=True)
    t.load_state_dict(teacher.state_dict())
    t.precision = prec
    t.to(dev)
    tr = NativeTrainer(m, t, n, (3, 160, 384), dev, phase=1, lr=1e-4)
    x = (rgb + noise).clamp(0, 1) if eps else rgb
    mean, mx, ymin = [], [], []
    for _ in range(steps):
        l = tr.step(x, speed, onehot, birdview=bv)
        mean.append(l.mean().item())
        mx.append(l.max().item())
        ymin.append(tr.last_pred[1][..., 1].min().item())
    res[name] = dict(mean=mean, max=mx, ymin=ymin)
    del tr
    if opt:
        _lib.check(_lib.get().lbc_config_set(opt[0].encode(), -1))
    print("%-13s first %.4f last %.4f max-of-means %.4f at step %d; min y %.4f" % (name, mean[0], mean[-1], max(mean), mean.index(max(mean)), min(ymin)), flush=True)
ref = torch.tensor(res["fp32"]["mean"])
for name in res:
    if name == "fp32":
        continue
    a = torch.tensor(res[name]["mean"])
    rel = (a - ref).abs() / ref.abs().clamp_min(1e-6)
    first_bad = int((rel > 0.10).nonzero()[0]) if (rel > 0.10).any() else -1
    print("%-13s vs fp32: max rel diff %.3f, first step beyond 10 %%: %d, median rel diff %.4f" % (name, rel.max(), first_bad, rel.median()))
json.dump(res, open(out, "w"))
