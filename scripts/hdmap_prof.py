"""Where the waves of the persistent halo-staged convolution spend their cycles (conv_hdmap.hpp, stamped build):
    python scripts/hdmap_prof.py [batch] [layer ...]      e.g.  python scripts/hdmap_prof.py 256 l2.conv l3.conv l4.conv
Per wave the kernel accumulates s_memtime deltas around the waits of every K-tile (LBC_HDMAP_PROF); printed: mean over waves of the
cycles per K-tile in (3 leading depth steps | vmcnt wait | lgkmcnt(0) + barrier | tail: reads + last depth step + DMA issue) and per
tile in the epilogue, next to the MFMA-bound cycles per K-tile (16 MFMAs x 32 cycles x 2 waves per SIMD = 1024)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learningbycheating_amd import _lib

SHAPES = {"l2.conv": (20, 48, 128, 128), "l3.conv": (10, 24, 256, 256), "l4.conv": (5, 12, 512, 512)}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layers = sys.argv[2:] or list(SHAPES)
lib = _lib.get()
dev = torch.device("cuda", 0)
P = _lib.ptr
for name in layers:
    H, W, C, K = SHAPES[name]
    x = torch.randn((N, H, W, C), device=dev).to(torch.bfloat16)
    w = (torch.randn((K, 3, 3, C), device=dev) * 0.05).to(torch.bfloat16)
    y = torch.empty((N, H, W, K), device=dev, dtype=torch.bfloat16)
    d = _lib.ConvDesc(N, H, W, C, K, 3, 3, 1, 1, 0, 3, 0)
    rows = ctypes.c_int(0)
    st = _lib.stream_for(x)
    run = lambda: _lib.check(lib.lbc_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, None, None, None, 0, P(y), None, ctypes.byref(rows), st))
    run(); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    plain_us = e0.elapsed_time(e1) * 100
    prof = torch.zeros((256, 8, 8), dtype=torch.int64, device=dev)
    _lib.config_set("LBC_HDMAP_PROF", prof.data_ptr())
    run(); torch.cuda.synchronize()
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    stamped_us = e0.elapsed_time(e1) * 1000
    _lib.config_set("LBC_HDMAP_PROF", -1)
    p = prof.cpu().double()
    live = p[:, :, 0] > 0
    q = p[live]                       # (waves, 8)
    kt, tiles = q[:, 0], q[:, 7]
    per = lambda i: (q[:, i] / kt).mean().item()
    print("%-8s batch %d: %.1f us plain, %.1f us stamped; %d waves, %.0f K-tiles and %.1f tiles per wave; s_memtime ticks per K-tile: steps %.0f | vmcnt %.0f | lgkm+barrier %.0f | tail %.0f "
          "= %.0f; epilogue %.0f per tile; whole stream %.0f ticks (max wave %.0f)"
          % (name, N, plain_us, stamped_us, q.shape[0], kt.mean().item(), tiles.mean().item(), per(1), per(2), per(3), per(4), per(1) + per(2) + per(3) + per(4),
             (q[:, 5] / tiles).mean().item(), q[:, 6].mean().item(), q[:, 6].max().item()), flush=True)
    # by wave index within the workgroup (arbitration order) and spread
    for wv in range(8):
        sel = p[:, wv][live[:, wv]]
        if len(sel):
            k2 = sel[:, 0]
            print("    wave %d: steps %.0f vmcnt %.0f barrier %.0f tail %.0f" % (wv, (sel[:, 1] / k2).mean(), (sel[:, 2] / k2).mean(), (sel[:, 3] / k2).mean(), (sel[:, 4] / k2).mean()))
