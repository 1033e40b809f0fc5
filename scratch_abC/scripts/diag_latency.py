"""GPU diagnostic: where does the per-tick time of the batch-1 inference session go (host copies, launch, wait mode)?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lbc_oracle as O
from learningbycheating_amd.bird_view.models import ImagePolicyModelSS
from learningbycheating_amd.inference import PolicySession
dev = torch.device("cuda", 0)
sd = O.make_state_dict("image", "resnet34", 51)
net = ImagePolicyModelSS("resnet34", all_branch=True); net.load_state_dict(sd)
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
net.precision = prec
ses = PolicySession(net, dev, use_graph=True)
frame = np.zeros((160, 384, 3), np.uint8)
def timeit(fn, n=100):
    for _ in range(10): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
print(prec, "run_step                       %.3f ms" % timeit(lambda: ses.run_step(frame, 3.0, 2)))
def replay_sync():
    ses.graph.replay(); torch.cuda.current_stream().synchronize()
print(prec, "graph.replay + stream sync     %.3f ms" % timeit(replay_sync))
ev = torch.cuda.Event()
def replay_spin():
    ses.graph.replay(); ev.record()
    while not ev.query(): pass
print(prec, "graph.replay + event spin      %.3f ms" % timeit(replay_spin))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ses.graph.replay()
e1.record(); torch.cuda.synchronize()
print(prec, "graph.replay back to back (GPU) %.3f ms" % (e0.elapsed_time(e1) / 50))
def host_only():
    f = torch.as_tensor(np.ascontiguousarray(frame)); ses.h_frame[0].copy_(f); ses.h_small.zero_()
print(prec, "host-side frame copy            %.3f ms" % timeit(host_only))
def h2d():
    ses.frame.copy_(ses.h_frame, non_blocking=True); torch.cuda.current_stream().synchronize()
print(prec, "H2D frame + sync                %.3f ms" % timeit(h2d))
def c1():
    ses.frame.copy_(ses.h_frame, non_blocking=True); ses.graph.replay(); torch.cuda.current_stream().synchronize()
print(prec, "H2D frame + replay + sync        %.3f ms" % timeit(c1))
def c2():
    ses.graph.replay(); ses.h_out.copy_(ses.out_sel[0], non_blocking=True); torch.cuda.current_stream().synchronize()
print(prec, "replay + D2H + sync              %.3f ms" % timeit(c2))
def c3():
    ses.speed.copy_(ses.h_small[:1], non_blocking=True); ses.command.copy_(ses.h_small[1:].view(1, 4), non_blocking=True); torch.cuda.current_stream().synchronize()
print(prec, "two small H2D + sync             %.3f ms" % timeit(c3))
def c4():
    ses.h_small.zero_(); ses.h_small[0] = 3.0; ses.h_small[2] = 1.0
print(prec, "h_small host writes              %.3f ms" % timeit(c4))
def c5():
    ses.frame.copy_(ses.h_frame, non_blocking=True); ses.speed.copy_(ses.h_small[:1], non_blocking=True); ses.command.copy_(ses.h_small[1:].view(1, 4), non_blocking=True)
    ses.graph.replay(); ses.h_out.copy_(ses.out_sel[0], non_blocking=True); torch.cuda.current_stream().synchronize()
print(prec, "all device work of run_step      %.3f ms" % timeit(c5))
def c6():
    return ses.h_out.numpy().copy()
print(prec, "h_out.numpy().copy()             %.3f ms" % timeit(c6))
for rep in range(3):
    print(prec, "run_step again (%d)              %.3f ms" % (rep, timeit(lambda: ses.run_step(frame, 3.0, 2))))
frame2 = np.random.randint(0, 256, (160, 384, 3), dtype=np.uint8)
print(prec, "run_step, random frame          %.3f ms" % timeit(lambda: ses.run_step(frame2, 3.0, 2)))
# line-by-line timing of run_step's body
import collections
acc = collections.OrderedDict()
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
self = ses
for it in range(60):
    t = time.perf_counter()
    f = torch.as_tensor(np.ascontiguousarray(frame2)); t = tick("as_tensor", t)
    ok = f.dtype != torch.uint8 or tuple(f.shape) != tuple(self.h_frame.shape[1:]); t = tick("check", t)
    self.h_frame[0].copy_(f); t = tick("h_frame.copy_", t)
    self.h_small.zero_(); self.h_small[0] = 3.0; self.h_small[2] = 1.0; t = tick("h_small", t)
    self.frame.copy_(self.h_frame, non_blocking=True); t = tick("H2D frame", t)
    self.speed.copy_(self.h_small[:1], non_blocking=True); t = tick("H2D speed", t)
    self.command.copy_(self.h_small[1:].view(1, 4), non_blocking=True); t = tick("H2D command", t)
    self.graph.replay(); t = tick("replay", t)
    self.h_out.copy_(self.out_sel[0], non_blocking=True); t = tick("D2H", t)
    torch.cuda.current_stream(self.device).synchronize(); t = tick("sync", t)
    r = self.h_out.numpy().copy(); t = tick("numpy", t)
for k, v in acc.items(): print("   %-14s %.3f ms" % (k, v / 60 * 1e3))
# alternatives for the frame upload
fr_t = torch.as_tensor(frame2)
def alt_pageable():
    self.frame[0].copy_(fr_t); self.graph.replay(); self.h_out.copy_(self.out_sel[0], non_blocking=True); torch.cuda.current_stream().synchronize()
print(prec, "pageable frame -> device, replay, D2H, sync   %.3f ms" % timeit(alt_pageable))
pins = [torch.zeros((1, 160, 384, 3), dtype=torch.uint8).pin_memory() for _ in range(4)]
cnt = [0]
def alt_ring():
    b = pins[cnt[0] & 3]; cnt[0] += 1
    b[0].copy_(fr_t); self.frame.copy_(b, non_blocking=True); self.graph.replay(); self.h_out.copy_(self.out_sel[0], non_blocking=True); torch.cuda.current_stream().synchronize()
print(prec, "ring of 4 pinned buffers                       %.3f ms" % timeit(alt_ring))
def alt_out_item():
    self.frame[0].copy_(fr_t); self.graph.replay(); r = self.out_sel[0].cpu()
print(prec, "pageable in, .cpu() out                        %.3f ms" % timeit(alt_out_item))
