import ctypes

import torch
import torch.nn.functional as F

from learningbycheating_amd import _lib


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def guarded(shape, device, fill=float("nan"), dtype=torch.float32):
    """tensor placed in the middle of a NaN-filled buffer: out-of-range writes/reads show up"""
    n = 1
    for s in shape:
        n *= s
    pad = 256
    buf = torch.full((n + 2 * pad,), fill, dtype=dtype, device=device)
    return buf, buf[pad:pad + n].view(shape)


def act_dtype(bf16):
    """element type of the activation tensors for lbc_conv_desc.bf16 (2 = bf16 tensors in HBM)"""
    return torch.bfloat16 if bf16 >= 2 else torch.float32


def check_guard(buf, n):
    pad = 256
    assert torch.isnan(buf[:pad]).all() and torch.isnan(buf[pad + n:]).all(), "kernel wrote outside its output"


class Conv:
    def __init__(self, device, split_floats=0):
        """split_floats > 0: every descriptor carries a split-K scratch of that many floats (lbc_conv_desc.split_workspace)"""
        self.dev = device
        self.lib = _lib.get()
        self.split_ws = torch.full((split_floats,), float("nan"), device=device) if split_floats else None

    def desc(self, N, H, W, C, K, k, s, p, relu=0, bf16=0, wt=0):
        d = _lib.ConvDesc(N, H, W, C, K, k, k, s, p, relu, bf16, wt)
        if self.split_ws is not None:
            d.split_workspace, d.split_workspace_bytes = _lib.ptr(self.split_ws), self.split_ws.numel() * 4
        return d

    def transpose(self, w3, A, T, B):
        """w3: device tensor [A][T][B] -> [B][T][A] through the library"""
        out = torch.empty((B, T, A), device=self.dev)
        _lib.check(self.lib.lbc_weight_transpose_f32(_lib.ptr(w3), _lib.ptr(out), A, T, B, _lib.stream_for(w3)))
        return out

    def fwd(self, x, w, stride, pad, bias=None, resid=None, pre=None, relu=0, stats=False, bf16=0):
        N, C, H, W = x.shape
        K, _, k, _ = w.shape
        d = self.desc(N, H, W, C, K, k, stride, pad, relu, bf16)
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        at = act_dtype(bf16)
        xh, wh = nhwc(x).to(self.dev).to(at), w.permute(0, 2, 3, 1).contiguous().to(self.dev)
        if bf16 == 3:
            wh = wh.to(torch.bfloat16)       # bf16 weight copy, same [K][kh][kw][C] layout
        buf, y = guarded((N, OH, OW, K), self.dev, dtype=at)
        rows = ctypes.c_int(0)
        keep = [t.to(self.dev) if t is not None else None for t in (bias, nhwc(resid).to(at) if resid is not None else None,
                                                                      pre[0] if pre else None, pre[1] if pre else None)]
        # (the row-count query describes the launch it is for: the same on-load transform and residual, NULL or not -- they select the kernel)
        _lib.check(self.lib.lbc_conv2d_fwd(ctypes.byref(d), None, None, None, _lib.ptr(keep[1]), _lib.ptr(keep[2]), _lib.ptr(keep[3]),
                                           1 if (pre and pre[2]) else 0, None, None, ctypes.byref(rows), None))
        st = torch.zeros((rows.value, 2, K), device=self.dev) if stats else None
        _lib.check(self.lib.lbc_conv2d_fwd(ctypes.byref(d), _lib.ptr(xh), _lib.ptr(wh), _lib.ptr(keep[0]), _lib.ptr(keep[1]),
                                           _lib.ptr(keep[2]), _lib.ptr(keep[3]), 1 if (pre and pre[2]) else 0, _lib.ptr(y),
                                           _lib.ptr(st), ctypes.byref(rows), _lib.stream_for(xh)))
        check_guard(buf, y.numel())
        return nchw(y).float().cpu(), (st.cpu() if stats else None)

    def dgrad(self, dy, w, H, W, stride, pad, resid=None, bf16=0, transposed=False):
        N, K = dy.shape[:2]
        _, C, k, _ = w.shape
        d = self.desc(N, H, W, C, K, k, stride, pad, 0, bf16, 1 if transposed else 0)
        at = act_dtype(bf16)
        dyh, wh = nhwc(dy).to(self.dev).to(at), w.permute(0, 2, 3, 1).contiguous().to(self.dev)
        if transposed:
            wh = self.transpose(wh.view(K, k * k, C), K, k * k, C)
        if bf16 == 3:
            wh = wh.to(torch.bfloat16)
        r = nhwc(resid).to(self.dev).to(at) if resid is not None else None
        buf, dx = guarded((N, H, W, C), self.dev, dtype=at)
        _lib.check(self.lib.lbc_conv2d_dgrad(ctypes.byref(d), _lib.ptr(dyh), _lib.ptr(wh), _lib.ptr(r), _lib.ptr(dx), _lib.stream_for(dyh)))
        check_guard(buf, dx.numel())
        return nchw(dx).float().cpu()

    def wgrad(self, x, dy, k, stride, pad, pre=None, beta=0.0, dw0=None, bf16=0):
        N, C, H, W = x.shape
        K = dy.shape[1]
        d = self.desc(N, H, W, C, K, k, stride, pad, 0, bf16)
        ws = torch.empty(self.lib.lbc_conv2d_wgrad_workspace(ctypes.byref(d)) // 4 + 1, device=self.dev)
        at = act_dtype(bf16)
        xh, dyh = nhwc(x).to(self.dev).to(at), nhwc(dy).to(self.dev).to(at)
        buf, dw = guarded((K, k, k, C), self.dev)
        if dw0 is not None:
            dw.copy_(dw0.permute(0, 2, 3, 1))
        keep = [pre[0].to(self.dev), pre[1].to(self.dev)] if pre else [None, None]
        _lib.check(self.lib.lbc_conv2d_wgrad(ctypes.byref(d), _lib.ptr(xh), _lib.ptr(dyh), _lib.ptr(keep[0]), _lib.ptr(keep[1]),
                                             1 if (pre and pre[2]) else 0, _lib.ptr(dw), beta, _lib.ptr(ws), _lib.stream_for(xh)))
        check_guard(buf, dw.numel())
        return dw.permute(0, 3, 1, 2).contiguous().cpu()

    def wgrad_group(self, xs, dys, pres=None, bf16=2):
        """lbc_conv2d_wgrad_group over len(xs) same-shaped 3x3 / stride-1 convolutions; pres: None or [(scale, shift)] per member (ReLU on)"""
        n = len(xs)
        N, C, H, W = xs[0].shape
        K = dys[0].shape[1]
        d = self.desc(N, H, W, C, K, 3, 1, 1, 0, bf16)
        assert self.lib.lbc_conv2d_wgrad_group_supported(ctypes.byref(d)) == 1
        ws = torch.empty(self.lib.lbc_conv2d_wgrad_group_workspace(ctypes.byref(d), n) // 4 + 1, device=self.dev)
        at = act_dtype(bf16)
        xh = [nhwc(x).to(self.dev).to(at) for x in xs]
        dyh = [nhwc(dy).to(self.dev).to(at) for dy in dys]
        outs = [guarded((K, 3, 3, C), self.dev) for _ in range(n)]
        ptrs = lambda ts: (ctypes.c_void_p * n)(*[_lib.ptr(t) for t in ts])
        keep = [[p[0].to(self.dev) for p in pres], [p[1].to(self.dev) for p in pres]] if pres else None
        _lib.check(self.lib.lbc_conv2d_wgrad_group(ctypes.byref(d), n, ptrs(xh), ptrs(dyh), ptrs(keep[0]) if pres else None,
                                                   ptrs(keep[1]) if pres else None, 1 if pres else 0, ptrs([o[1] for o in outs]),
                                                   _lib.ptr(ws), _lib.stream_for(xh[0])))
        for buf, dw in outs:
            check_guard(buf, dw.numel())
        return [dw.permute(0, 3, 1, 2).contiguous().cpu() for _, dw in outs]

    def deconv_all(self, x, w, bias, pre, relu, bf16=0):
        """fwd, dgrad and wgrad of ConvTranspose2d(k3,s2,p1,op1) with BN-on-load; returns (y, stats, fn(dy)->(dx, dw))"""
        N, C, H, W = x.shape
        K = w.shape[1]
        d = self.desc(N, H, W, C, K, 3, 2, 1, relu, bf16)
        dfwd = self.desc(N, H, W, C, K, 3, 2, 1, relu, bf16, 1 if bf16 else 0)
        at = act_dtype(bf16)
        xh, wh = nhwc(x).to(self.dev).to(at), w.permute(0, 2, 3, 1).contiguous().to(self.dev)
        wfwd = self.transpose(wh.view(C, 9, K), C, 9, K) if bf16 else wh
        ps, pt, b = pre[0].to(self.dev), pre[1].to(self.dev), bias.to(self.dev)
        rows = ctypes.c_int(0)
        _lib.check(self.lib.lbc_deconv3x3s2_fwd(ctypes.byref(dfwd), None, None, None, None, None, 0, None, None, ctypes.byref(rows), None))
        st = torch.zeros((rows.value, 2, K), device=self.dev)
        buf, y = guarded((N, 2 * H, 2 * W, K), self.dev, dtype=at)
        _lib.check(self.lib.lbc_deconv3x3s2_fwd(ctypes.byref(dfwd), _lib.ptr(xh), _lib.ptr(wfwd), _lib.ptr(b), _lib.ptr(ps), _lib.ptr(pt), 0,
                                                _lib.ptr(y), _lib.ptr(st), ctypes.byref(rows), _lib.stream_for(xh)))
        check_guard(buf, y.numel())

        def bwd(dy):
            dyh = nhwc(dy).to(self.dev).to(at)
            bufx, dx = guarded((N, H, W, C), self.dev, dtype=at)
            _lib.check(self.lib.lbc_deconv3x3s2_dgrad(ctypes.byref(d), _lib.ptr(dyh), _lib.ptr(wh), _lib.ptr(dx), _lib.stream_for(dyh)))
            check_guard(bufx, dx.numel())
            ws = torch.empty(self.lib.lbc_deconv3x3s2_wgrad_workspace(ctypes.byref(d)) // 4 + 1, device=self.dev)
            bufw, dw = guarded((C, 3, 3, K), self.dev)
            _lib.check(self.lib.lbc_deconv3x3s2_wgrad(ctypes.byref(d), _lib.ptr(xh), _lib.ptr(dyh), _lib.ptr(ps), _lib.ptr(pt), 0,
                                                      _lib.ptr(dw), 0.0, _lib.ptr(ws), _lib.stream_for(xh)))
            check_guard(bufw, dw.numel())
            return nchw(dx).float().cpu(), dw.permute(0, 3, 1, 2).contiguous().cpu()
        return nchw(y).float().cpu(), st.cpu(), bwd


def relerr(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def engine_from_state_dict(sd, kind, backbone, H, W, max_batch, device, precision=0):
    from learningbycheating_amd.engine import PolicyEngine
    eng = PolicyEngine(34 if backbone == "resnet34" else 18, 3 if kind == "image" else 7, H, W, kind == "image", max_batch, device, precision)
    tens = {k: (v.to(device).contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v.clone().to(device))
            for k, v in sd.items() if k in set(eng.names)}
    eng.bind(tens, True)
    return eng, tens
