"""GPU diagnostic: where does a hipGraph replay of the batch-1 eval forward diverge from eager launches?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lbc_oracle as O
from learningbycheating_amd.bird_view.models import ImagePolicyModelSS
from learningbycheating_amd.inference import PolicySession
dev = torch.device("cuda", 0)
sd = O.make_state_dict("image", "resnet18", 51)
outs = {}
for mode in ("eager", "graph"):
    net = ImagePolicyModelSS("resnet18", all_branch=True); net.load_state_dict(sd)
    ses = PolicySession(net, dev, use_graph=(mode == "graph"))
    g = torch.Generator().manual_seed(1)
    frame = torch.randint(0, 256, (160, 384, 3), generator=g, dtype=torch.uint8).numpy()
    r = ses.run_step(frame, 3.0, 2)
    torch.cuda.synchronize()
    ws = ses.eng.workspace.view(torch.float32).float().nan_to_num(0.0, 0.0, 0.0)
    chunks = ws[: ws.numel() // 65536 * 65536].view(-1, 65536).double().abs().sum(1).cpu()
    outs[mode] = (r, chunks, ses.frame.clone().cpu(), ses.command.clone().cpu(), ses.speed.clone().cpu())
    print(mode, r.reshape(-1)[:4], "frame sum", int(ses.frame.long().sum()), "cmd", ses.command.cpu().tolist(), "speed", ses.speed.item())
a, b = outs["eager"][1], outs["graph"][1]
bad = (a - b).abs() > 1e-6 * (a.abs() + 1)
print("differing 256KB workspace chunks:", int(bad.sum()), "of", len(a), "first:", bad.nonzero().reshape(-1)[:12].tolist())
