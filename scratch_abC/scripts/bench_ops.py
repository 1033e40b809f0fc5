"""Per-layer timing of the convolution family through the C ABI (include/lbc_hip.h) on the ResNet-34 / decoder shapes
of ImagePolicyModelSS at a given batch.  Prints ms and TFLOP/s per (layer, op, mode); mode 0 = exact f32, 1 = bf16 MFMA
operands with f32 tensors, 2 = bf16 operands and bf16 tensors, 3 = 2 + bf16 weight copies (fwd / dgrad).  Usage: python scripts/bench_ops.py [batch] [modes] [ops]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from learningbycheating_amd import _lib

LAYERS = [  # name, H, W, C, K, k, s, p     (input geometry; 160x384 RGB network)
    ("l1.conv", 40, 96, 64, 64, 3, 1, 1),
    ("l2.0.c1", 40, 96, 64, 128, 3, 2, 1),
    ("l2.conv", 20, 48, 128, 128, 3, 1, 1),
    ("l2.ds", 40, 96, 64, 128, 1, 2, 0),
    ("l3.0.c1", 20, 48, 128, 256, 3, 2, 1),
    ("l3.conv", 10, 24, 256, 256, 3, 1, 1),
    ("l4.0.c1", 10, 24, 256, 512, 3, 2, 1),
    ("l4.conv", 5, 12, 512, 512, 3, 1, 1),
]
DECONVS = [("dec0", 5, 12, 640, 256), ("dec1", 10, 24, 256, 128), ("dec2", 20, 48, 128, 64)]


def timeit(fn, iters=int(os.environ.get("BENCH_OPS_ITERS", "10"))):
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
    modes = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "012")]
    ops = (sys.argv[3] if len(sys.argv) > 3 else "fwd,dgrad,wgrad").split(",")
    only = sys.argv[4] if len(sys.argv) > 4 else ""          # substring filter on the layer name
    lib = _lib.get()
    dev = torch.device("cuda", 0)
    P = _lib.ptr
    st = None
    print("%-8s %-6s %4s %9s %9s" % ("layer", "op", "mode", "ms", "TFLOP/s"))
    for name, H, W, C, K, k, s, p in LAYERS:
        if only and only not in name:
            continue
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        flops = 2.0 * N * OH * OW * K * C * k * k
        for mode in modes:
            at = torch.bfloat16 if mode >= 2 else torch.float32
            x = torch.randn((N, H, W, C), device=dev).to(at)
            dy = torch.randn((N, OH, OW, K), device=dev).to(at)
            w = torch.randn((K, k, k, C), device=dev) * 0.05
            # BENCH_OPS_FILL: zero = all-zero operands, relu = activations post-ReLU (half zeros): the chip clocks to its power budget,
            # so operand values move the launch time (MI355X_MICROARCH.md "DVFS give-back")
            fill = os.environ.get("BENCH_OPS_FILL", "")
            if fill == "zero":
                x, dy, w = x * 0, dy * 0, w * 0
            elif fill == "relu":
                x = torch.relu(x)
            wt = torch.empty((C, k * k, K), device=dev)
            y = torch.empty((N, OH, OW, K), device=dev, dtype=at)
            dx = torch.empty((N, H, W, C), device=dev, dtype=at)
            dw = torch.empty_like(w)
            ps, pt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
            d = _lib.ConvDesc(N, H, W, C, K, k, k, s, p, 0, mode, 0)
            dT = _lib.ConvDesc(N, H, W, C, K, k, k, s, p, 0, mode, 1)
            if os.environ.get("BENCH_OPS_SPLIT_WS"):     # split-K scratch (lbc_conv_desc.split_workspace): what the executor hands its launches
                sws = torch.empty(8 * N * H * W * max(C, K), device=dev)
                for dd in (d, dT):
                    dd.split_workspace, dd.split_workspace_bytes = P(sws), sws.numel() * 4
            st = _lib.stream_for(x)
            _lib.check(lib.lbc_weight_transpose_f32(P(w), P(wt), K, k * k, C, st))
            ws = torch.empty(lib.lbc_conv2d_wgrad_workspace(ctypes.byref(d)) // 4 + 1, device=dev)
            rows = ctypes.c_int(0)
            if mode == 3:          # bf16 weight copies (what the executor's weight_prep produces)
                w, wt = w.to(torch.bfloat16), wt.to(torch.bfloat16)
            runs = {
                "fwd": lambda: _lib.check(lib.lbc_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, None, None, None, 0, P(y), None, ctypes.byref(rows), st)),
                "fwd+bn": lambda: _lib.check(lib.lbc_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, None, P(ps), P(pt), 1, P(y), None, ctypes.byref(rows), st)),
                "dgrad": lambda: _lib.check(lib.lbc_conv2d_dgrad(ctypes.byref(dT if mode else d), P(dy), P(wt if mode else w), None, P(dx), st)),
                "wgrad": lambda: _lib.check(lib.lbc_conv2d_wgrad(ctypes.byref(d), P(x), P(dy), None, None, 0, P(dw), 0.0, P(ws), st)),
                "wgrad+bn": lambda: _lib.check(lib.lbc_conv2d_wgrad(ctypes.byref(d), P(x), P(dy), P(ps), P(pt), 1, P(dw), 0.0, P(ws), st)),
            }
            for op in ops:
                if op not in runs:
                    continue
                ms = timeit(runs[op])
                print("%-8s %-8s %2d %9.3f %9.1f" % (name, op, mode, ms, flops / ms * 1e-9), flush=True)
    if "deconv" in ops or len(sys.argv) <= 3:
        for name, H, W, C, K in DECONVS:
            flops = 2.0 * N * H * W * C * K * 9
            for mode in modes:
                at = torch.bfloat16 if mode == 2 else torch.float32
                x = torch.randn((N, H, W, C), device=dev).to(at)
                dy = torch.randn((N, 2 * H, 2 * W, K), device=dev).to(at)
                w = torch.randn((C, 3, 3, K), device=dev) * 0.05
                wt = torch.empty((K, 9, C), device=dev)
                b = torch.randn(K, device=dev)
                y = torch.empty((N, 2 * H, 2 * W, K), device=dev, dtype=at)
                dx = torch.empty((N, H, W, C), device=dev, dtype=at)
                dw = torch.empty_like(w)
                ps, pt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
                d = _lib.ConvDesc(N, H, W, C, K, 3, 3, 2, 1, 1, mode, 0)
                dT = _lib.ConvDesc(N, H, W, C, K, 3, 3, 2, 1, 1, mode, 1)
                st = _lib.stream_for(x)
                _lib.check(lib.lbc_weight_transpose_f32(P(w), P(wt), C, 9, K, st))
                ws = torch.empty(lib.lbc_deconv3x3s2_wgrad_workspace(ctypes.byref(d)) // 4 + 1, device=dev)
                rows = ctypes.c_int(0)
                runs = {
                    "fwd": lambda: _lib.check(lib.lbc_deconv3x3s2_fwd(ctypes.byref(dT if mode else d), P(x), P(wt if mode else w), P(b), P(ps), P(pt), 0, P(y), None, ctypes.byref(rows), st)),
                    "dgrad": lambda: _lib.check(lib.lbc_deconv3x3s2_dgrad(ctypes.byref(d), P(dy), P(w), P(dx), st)),
                    "wgrad": lambda: _lib.check(lib.lbc_deconv3x3s2_wgrad(ctypes.byref(d), P(x), P(dy), P(ps), P(pt), 0, P(dw), 0.0, P(ws), st)),
                }
                for op in ("fwd", "dgrad", "wgrad"):
                    ms = timeit(runs[op])
                    print("%-8s %-8s %2d %9.3f %9.1f" % (name, op, mode, ms, flops / ms * 1e-9), flush=True)


if __name__ == "__main__":
    main()
