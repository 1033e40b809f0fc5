"""GPU diagnostic: per-tensor gradient error of the executor vs the fp64 oracle, listed in backward order."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lbc_oracle as O
from tests.helpers import engine_from_state_dict, relerr
import tests.test_model as T

kind, backbone, h, w, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
TD = torch.float32 if (len(sys.argv) > 6 and sys.argv[6] == "f32") else torch.float64     # truth dtype (f64 needs ~0.4 GB / image)
dev = torch.device("cuda", 0)
sd = O.make_state_dict(kind, backbone, 3, h, w)
x, speed, cmd = T._inputs(kind, n, h, w, 4)
eng, tens = engine_from_state_dict(sd, kind, backbone, h, w, n, dev)
ps, pa = eng.forward(x.to(dev), speed.to(dev), cmd.to(dev), True)
g = torch.Generator().manual_seed(5)
d_all, d_sel = torch.randn((n, 4, 5, 2), generator=g), torch.randn((n, 5, 2), generator=g)
eng.backward(d_sel.to(dev), d_all.to(dev))
torch.cuda.synchronize()
sd64 = {k: (v.to(TD) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
sp64 = O.as_params(sd64)
o64s, o64a = O.policy_forward(sp64, kind, backbone, x.to(TD), speed.to(TD), cmd.to(TD), True)
((o64a * d_all.to(TD)).sum() + (o64s * d_sel.to(TD)).sum()).backward()
sp32 = O.as_params(sd)
o32s, o32a = O.policy_forward(sp32, kind, backbone, x, speed, cmd, True)
((o32a * d_all).sum() + (o32s * d_sel).sum()).backward()
print("env", {k: v for k, v in os.environ.items() if k.startswith("LBC_")})
print("fwd train: hip-truth %.2e  torch32-truth %.2e" % ((pa.cpu().double() - o64a.double()).abs().max().item(), (o32a.double() - o64a.double()).abs().max().item()))
names = list(eng.grad_views.keys())[::-1]
for k in names:
    if k.startswith("location_pred") and k.endswith("bias"):
        continue
    e = relerr(eng.grad_views[k].cpu().double(), sp64[k].grad.double())
    e32 = relerr(sp32[k].grad.double(), sp64[k].grad.double())
    flag = " <<<" if e > 20 * max(e32, 1e-6) else ""
    cos = torch.nn.functional.cosine_similarity(eng.grad_views[k].cpu().double().reshape(1, -1), sp64[k].grad.double().reshape(1, -1)).item()
    print("%-44s hip %.2e  torch32 %.2e  max|g| %.2e  cos %.6f%s" % (k, e, e32, sp64[k].grad.abs().max().item(), cos, flag))
