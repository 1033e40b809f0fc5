"""Where the waves of the 64-channel persistent convolution spend their time (conv_c64p.hip with LBC_C64P_PROF stamps):
    python scripts/c64p_prof.py [batch]
Per wave the kernel accumulates s_memtime deltas per tile in: waiting for the tile's halo + the opening barrier | the on-load
BatchNorm transform (PRE form) | the 72-MFMA K loop | the wave-private epilogue.  Printed: the mean share of each phase, the
K-loop ticks per tile next to its MFMA-bound length, for the plain forward, forward + BatchNorm-on-load, forward + residual,
and the input gradient (plain / + residual)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learningbycheating_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W, C, K = 40, 96, 64, 64
lib = _lib.get()
dev = torch.device("cuda", 0)
P = _lib.ptr
x = torch.randn((N, H, W, C), device=dev).to(torch.bfloat16)
r = torch.randn((N, H, W, K), device=dev).to(torch.bfloat16)
w = (torch.randn((K, 3, 3, C), device=dev) * 0.05).to(torch.bfloat16)
wt = w.permute(3, 1, 2, 0).contiguous()
y = torch.empty((N, H, W, K), device=dev, dtype=torch.bfloat16)
ps, pt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
d = _lib.ConvDesc(N, H, W, C, K, 3, 3, 1, 1, 0, 3, 0)
dt = _lib.ConvDesc(N, H, W, C, K, 3, 3, 1, 1, 0, 3, 1)
rows = ctypes.c_int(0)
st = _lib.stream_for(x)
stats = torch.zeros((4096, 2, K), device=dev)
CASES = {
    "fwd plain + stats": lambda: _lib.check(lib.lbc_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, None, None, None, 0, P(y), P(stats), ctypes.byref(rows), st)),
    "fwd BatchNorm-on-load": lambda: _lib.check(lib.lbc_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, None, P(ps), P(pt), 1, P(y), P(stats), ctypes.byref(rows), st)),
    "fwd + residual": lambda: _lib.check(lib.lbc_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, P(r), None, None, 0, P(y), None, ctypes.byref(rows), st)),
    "dgrad plain": lambda: _lib.check(lib.lbc_conv2d_dgrad(ctypes.byref(dt), P(x), P(wt), None, P(y), st)),
    "dgrad + residual": lambda: _lib.check(lib.lbc_conv2d_dgrad(ctypes.byref(dt), P(x), P(wt), P(r), P(y), st)),
}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, run in CASES.items():
    run(); run(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    plain_us = e0.elapsed_time(e1) * 100
    prof = torch.zeros((1024, 8, 8), dtype=torch.int64, device=dev)
    _lib.config_set("LBC_C64P_PROF", prof.data_ptr())
    run(); torch.cuda.synchronize()
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    stamped_us = e0.elapsed_time(e1) * 1000
    _lib.config_set("LBC_C64P_PROF", -1)
    p = prof.cpu().double().reshape(-1, 8)          # [workgroup * waves per workgroup + wave][8] (8 or 4 waves per workgroup)
    q = p[p[:, 0] > 0]
    nw = 8 if int(_lib.config_get("LBC_C64P_BM")) == 256 else 4
    p = p[: (p.shape[0] // nw) * nw].reshape(-1, nw, 8)
    tiles = q[:, 0]
    tot = q[:, 5]
    sh = lambda i: (q[:, i] / tot).mean().item()
    ticks_per_us = tot.max().item() / stamped_us
    print("%-22s batch %d: %.1f us plain, %.1f us stamped; %d waves x %.1f tiles; share of a wave's time: halo wait + barrier %.2f | on-load transform %.2f | K loop %.2f | "
          "epilogue %.2f | rest %.2f; per tile: K loop %.2f us (MFMA-bound: 72 MFMAs x 32 cycles x 2 waves per SIMD = 4608 cycles = %.2f us at 2.4 GHz), wait %.2f us, epilogue %.2f us"
          % (name, N, plain_us, stamped_us, q.shape[0], tiles.mean().item(), sh(1), sh(2), sh(3), sh(4), 1 - sh(1) - sh(2) - sh(3) - sh(4),
             (q[:, 3] / tiles).mean().item() / ticks_per_us, 4608 / 2400.0, (q[:, 1] / tiles).mean().item() / ticks_per_us, (q[:, 4] / tiles).mean().item() / ticks_per_us), flush=True)
    for wv in range(nw):
        sel = p[:, wv][p[:, wv, 0] > 0]
        if len(sel):
            print("    wave %d: wait %.2f pre %.2f K %.2f epi %.2f" % ((wv,) + tuple((sel[:, i] / sel[:, 5]).mean().item() for i in (1, 2, 3, 4))))
