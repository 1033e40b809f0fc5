"""GPU diagnostic: f32 vs bf16 phase-1 training from a common warm start (loss curves, first-step gradient agreement)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lbc_oracle as O
from oracle.make_golden import seeded_inputs
from learningbycheating_amd.bird_view.models import ImagePolicyModelSS, BirdViewPolicyModelSS
from learningbycheating_amd.training.native import NativeTrainer
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rgb, speed, cmd = seeded_inputs("image", n, 41); bv, _, _ = seeded_inputs("birdview", n, 42)
onehot = O.one_hot(cmd).to(dev); rgb, speed, bv = rgb.to(dev), speed.to(dev), bv.to(dev)
g = torch.Generator().manual_seed(43)
tgt = torch.rand((n, 4, 5, 2), generator=g); tgt[..., 0] = tgt[..., 0] * 1.2 - 0.6; tgt[..., 1] = tgt[..., 1] * 0.5 + 0.3
torch.manual_seed(44); student = ImagePolicyModelSS("resnet34", all_branch=True).to(dev)
torch.manual_seed(45); teacher = BirdViewPolicyModelSS("resnet18", all_branch=True).to(dev)
warm = NativeTrainer(student, None, n, (3, 160, 384), dev, phase="l1_all", lr=1e-3)
for _ in range(40): warm.step(rgb, speed, onehot, target=tgt.to(dev))
torch.cuda.synchronize(); del warm
ckpt = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
res = {}
for prec in ("fp32", "bf16"):
    m = ImagePolicyModelSS("resnet34", all_branch=True); m.load_state_dict(ckpt); m.precision = prec; m = m.to(dev)
    t = BirdViewPolicyModelSS("resnet18", all_branch=True); t.load_state_dict(teacher.state_dict()); t.precision = prec; t.to(dev)
    tr = NativeTrainer(m, t, n, (3, 160, 384), dev, phase=1, lr=1e-4)
    l0 = tr.step(rgb, speed, onehot, birdview=bv, update=False).clone()
    for st in range(tr.nstages): tr.eng.backward(None, tr.dpred_all[:n], st)
    torch.cuda.synchronize()
    grads = {k: v.clone().cpu() for k, v in tr.eng.grad_views.items()}
    pred = tr.last_pred[1].clone().cpu()
    curve = [tr.step(rgb, speed, onehot, birdview=bv).mean().item() for _ in range(50)]
    res[prec] = (l0.cpu(), grads, curve, pred)
    print(prec, "per-sample loss first step: min %.3f max %.3f mean %.3f; min pred y %.4f" % (l0.min(), l0.max(), l0.mean(), pred[..., 1].min()))
    print(prec, "curve", " ".join("%.3f" % c for c in curve))
ga, gb = res["fp32"][1], res["bf16"][1]
cos = sorted((torch.nn.functional.cosine_similarity(ga[k].reshape(1, -1).double(), gb[k].reshape(1, -1).double()).item(), k) for k in ga if ga[k].norm() > 1e-9)
print("first-step gradient cosine bf16 vs f32: min %.4f (%s) p10 %.4f median %.4f" % (cos[0][0], cos[0][1], cos[len(cos) // 10][0], cos[len(cos) // 2][0]))
print("worst 6:", [(round(c, 3), k) for c, k in cos[:6]])
print("|pred bf16 - f32| max", (res["bf16"][3] - res["fp32"][3]).abs().max().item())
