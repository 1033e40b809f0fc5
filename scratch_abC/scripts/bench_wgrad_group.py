"""Grouped weight-gradient launch (lbc_conv2d_wgrad_group) on the four ResNet-34 stage shapes:  python scripts/bench_wgrad_group.py [batch]
Prints ms and TFLOP/s per stage for the group sizes the training step uses, next to n single launches of lbc_conv2d_wgrad."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from learningbycheating_amd import _lib

STAGES = [("layer1", 40, 96, 64, 6), ("layer2", 20, 48, 128, 7), ("layer3", 10, 24, 256, 11), ("layer4", 5, 12, 512, 5)]


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lib = _lib.get()
    dev = torch.device("cuda", 0)
    P = _lib.ptr
    for name, H, W, C, n in STAGES:
        d = _lib.ConvDesc(N, H, W, C, C, 3, 3, 1, 1, 0, 2, 0)
        xs = [torch.randn((N, H, W, C), device=dev).to(torch.bfloat16) for _ in range(n)]
        dys = [torch.randn((N, H, W, C), device=dev).to(torch.bfloat16) for _ in range(n)]
        dws = [torch.empty((C, 3, 3, C), device=dev) for _ in range(n)]
        st = _lib.stream_for(xs[0])
        wsb = lib.lbc_conv2d_wgrad_group_workspace(ctypes.byref(d), n)
        ws = torch.empty(wsb // 4 + 1, device=dev)
        ws1 = torch.empty(lib.lbc_conv2d_wgrad_workspace(ctypes.byref(d)) // 4 + 1, device=dev)
        arr = lambda ts: (ctypes.c_void_p * n)(*[P(t) for t in ts])
        ax, ady, adw = arr(xs), arr(dys), arr(dws)
        flops = 2.0 * N * H * W * C * C * 9 * n
        tg = timeit(lambda: _lib.check(lib.lbc_conv2d_wgrad_group(ctypes.byref(d), n, ax, ady, None, None, 0, adw, P(ws), st)))

        def singles():
            for i in range(n):
                _lib.check(lib.lbc_conv2d_wgrad(ctypes.byref(d), P(xs[i]), P(dys[i]), None, None, 0, P(dws[i]), 0.0, P(ws1), st))
        t1 = timeit(singles)
        print("%-7s n=%2d  group %.3f ms %7.1f TF/s (slabs %.0f MB)   %d single launches %.3f ms %7.1f TF/s"
              % (name, n, tg, flops / tg / 1e9, wsb / 1e6, n, t1, flops / t1 / 1e9))


if __name__ == "__main__":
    main()
