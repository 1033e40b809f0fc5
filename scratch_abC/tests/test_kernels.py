"""Per-kernel parity of the convolution family against torch CPU (conv2d / conv_transpose2d + autograd).
Small shapes run the kernel sources under the CPU emulator (tests/emu); the gpu-marked cases run the
real gfx950 library at the layer shapes of SURVEY.md appendix B.1 through the C ABI."""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import Conv, relerr

gpu = pytest.mark.gpu
# (N, H, W, Cin, Cout, k, stride, pad)
SMALL = [(1, 8, 8, 64, 64, 3, 1, 1), (2, 5, 12, 64, 128, 3, 1, 1), (1, 10, 12, 64, 128, 3, 2, 1), (1, 8, 12, 64, 128, 1, 2, 0),
         (3, 6, 7, 64, 64, 3, 1, 1)]
REAL = [pytest.param(c, marks=gpu) for c in [
    (2, 40, 96, 64, 64, 3, 1, 1), (32, 40, 96, 64, 64, 3, 1, 1),        # layer1
    (2, 40, 96, 64, 128, 3, 2, 1), (2, 40, 96, 64, 128, 1, 2, 0),        # layer2.0 conv1 / downsample
    (8, 20, 48, 128, 128, 3, 1, 1),                                      # layer2
    (2, 20, 48, 128, 256, 3, 2, 1), (8, 10, 24, 256, 256, 3, 1, 1),      # layer3
    (2, 10, 24, 256, 512, 3, 2, 1), (32, 5, 12, 512, 512, 3, 1, 1), (2, 10, 24, 256, 512, 1, 2, 0),   # layer4
    (3, 6, 6, 512, 512, 3, 1, 1),                                        # bird-view layer4 (6x6)
    # BASELINE config 2's batch (64): long reductions, many split-K slabs, 128-row tiles
    (64, 40, 96, 64, 64, 3, 1, 1), (64, 20, 48, 128, 128, 3, 1, 1), (64, 10, 24, 256, 256, 3, 1, 1), (64, 5, 12, 512, 512, 3, 1, 1),
    (64, 40, 96, 64, 128, 3, 2, 1), (64, 20, 48, 128, 256, 1, 2, 0)]]


def make(cfg, seed=0):
    N, H, W, C, K, k, s, p = cfg
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, C, H, W), generator=g)
    w = torch.randn((K, C, k, k), generator=g) * (2.0 / (C * k * k)) ** 0.5
    return x, w


@pytest.mark.parametrize("cfg", SMALL + [(2, 5, 5, 32, 64, 3, 1, 1)] + REAL)
def test_conv_fwd(env, cfg):
    dev, _ = env
    N, H, W, C, K, k, s, p = cfg
    x, w = make(cfg)
    ref = F.conv2d(x, w, None, s, p)
    y, st = Conv(dev).fwd(x, w, s, p, stats=True)
    assert relerr(y, ref) < 1e-5
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-3 * ref.abs().sum((0, 2, 3)).max().item() / 1e2)
    assert torch.allclose(st[:, 1].sum(0), (ref * ref).sum((0, 2, 3)), rtol=1e-4)


@pytest.mark.parametrize("cfg", [SMALL[1], SMALL[2]] + [pytest.param((4, 20, 48, 128, 128, 3, 1, 1), marks=gpu)])
def test_conv_fwd_fused_prologue_epilogue(env, cfg):
    dev, _ = env
    N, H, W, C, K, k, s, p = cfg
    x, w = make(cfg, 1)
    g = torch.Generator().manual_seed(2)
    ps, pt, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g), torch.randn(K, generator=g)
    xin = F.relu(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
    ref = F.conv2d(xin, w, b, s, p)
    r = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref + r)
    y, st = Conv(dev).fwd(x, w, s, p, bias=b, resid=r, pre=(ps, pt, True), relu=1, stats=True)
    assert relerr(y, ref) < 1e-5
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("cfg", SMALL + REAL)
def test_conv_dgrad(env, cfg):
    dev, _ = env
    N, H, W, C, K, k, s, p = cfg
    if k == 1 and s == 2:
        pytest.skip("1x1/2 dgrad is exercised with accumulation in test_conv_dgrad_residual")
    x, w = make(cfg, 3)
    x.requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(4))
    y.backward(dy)
    dx = Conv(dev).dgrad(dy, w, H, W, s, p)
    assert relerr(dx, x.grad) < 1e-5


@pytest.mark.parametrize("cfg", [SMALL[1], SMALL[3]] + [pytest.param((2, 40, 96, 64, 128, 1, 2, 0), marks=gpu)])
def test_conv_dgrad_residual(env, cfg):
    dev, _ = env
    N, H, W, C, K, k, s, p = cfg
    x, w = make(cfg, 5)
    x.requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(6))
    y.backward(dy)
    r = torch.randn(x.shape, generator=torch.Generator().manual_seed(7))
    dx = Conv(dev).dgrad(dy, w, H, W, s, p, resid=r)
    assert relerr(dx, x.grad + r) < 1e-5


@pytest.mark.parametrize("cfg", SMALL + [(40, 5, 6, 64, 64, 3, 1, 1)] + REAL)
def test_conv_wgrad(env, cfg):
    dev, _ = env
    N, H, W, C, K, k, s, p = cfg
    x, w = make(cfg, 8)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9))
    y.backward(dy)
    dw = Conv(dev).wgrad(x, dy, k, s, p)
    assert relerr(dw, w.grad) < 2e-5


def test_conv_wgrad_fused_bn_on_load_and_accumulate(env):
    dev, _ = env
    cfg = (3, 9, 11, 64, 128, 3, 1, 1)
    N, H, W, C, K, k, s, p = cfg
    x, w = make(cfg, 10)
    g = torch.Generator().manual_seed(11)
    ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    w.requires_grad_(True)
    y = F.conv2d(F.relu(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1)), w, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw0 = torch.randn(w.shape, generator=g)
    dw = Conv(dev).wgrad(x, dy, k, s, p, pre=(ps, pt, True), beta=1.0, dw0=dw0)
    assert relerr(dw, w.grad + dw0) < 2e-5


DEC_SMALL = [(2, 3, 4, 64, 64), (1, 5, 12, 128, 64)]
DEC_REAL = [pytest.param(c, marks=gpu) for c in [(4, 5, 12, 640, 256), (4, 10, 24, 256, 128), (2, 20, 48, 128, 64), (2, 6, 6, 640, 256),
                                                  (64, 5, 12, 640, 256), (64, 10, 24, 256, 128), (64, 20, 48, 128, 64)]]   # + BASELINE config 2's batch


@pytest.mark.parametrize("cfg", DEC_SMALL + DEC_REAL)
def test_deconv_fwd_dgrad_wgrad(env, cfg):
    """ConvTranspose2d(k3,s2,p1,op1) with the preceding BatchNorm applied on load, bias + ReLU + statistics fused"""
    dev, _ = env
    N, H, W, C, K = cfg
    g = torch.Generator().manual_seed(12)
    x = torch.randn((N, C, H, W), generator=g)
    w = (torch.randn((C, K, 3, 3), generator=g) * (2.0 / (C * 2.25)) ** 0.5).requires_grad_(True)
    b = torch.randn(K, generator=g)
    ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xn = (x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1)).requires_grad_(True)
    u = F.conv_transpose2d(xn, w, b, 2, 1, 1)
    ref = F.relu(u)
    y, st, bwd = Conv(dev).deconv_all(x, w.detach(), b, (ps, pt), relu=1)
    assert relerr(y, ref) < 1e-5
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    dy = torch.randn(u.shape, generator=g)
    u.backward(dy)
    dx, dw = bwd(dy)
    assert relerr(dx, xn.grad) < 1e-5 and relerr(dw, w.grad) < 2e-5


# ---- bf16-MFMA compute mode: operands rounded to bf16 (RNE) inside the kernel, f32 accumulation and storage --------
def rbf(t):
    return t.to(torch.bfloat16).to(torch.float32)


BF_SMALL = [(1, 8, 8, 64, 64, 3, 1, 1), (2, 5, 12, 64, 128, 3, 1, 1), (1, 10, 12, 128, 128, 3, 2, 1), (1, 8, 12, 64, 128, 1, 2, 0)]
BF_REAL = [pytest.param(c, marks=gpu) for c in [(8, 40, 96, 64, 64, 3, 1, 1), (8, 20, 48, 128, 128, 3, 1, 1), (2, 20, 48, 128, 256, 3, 2, 1),
                                                 (16, 5, 12, 512, 512, 3, 1, 1), (2, 10, 24, 256, 512, 1, 2, 0)]]


# mode 1: f32 tensors, operands rounded inside the kernel.  mode 2: the activation tensors themselves are bf16 in HBM
# (inputs pre-rounded, outputs rounded once on store -> half a bf16 ulp = 2^-9 relative on top of mode 1's error)
MODES = [1, 2]
OUT_TOL = {1: 0.0, 2: 2.0 ** -8}


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cfg", BF_SMALL + BF_REAL)
def test_conv_fwd_bf16_mode(env, cfg, mode):
    """against an f32 convolution of the bf16-rounded operands (tight), and against the unrounded one (bf16-level)"""
    dev, _ = env
    N, H, W, C, K, k, s, p = cfg
    x, w = make(cfg, 20)
    if mode == 2:
        x = rbf(x)
    g = torch.Generator().manual_seed(21)
    ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xin = F.relu(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))      # BN+ReLU on load happens in f32, before the rounding
    ref = F.conv2d(rbf(xin), rbf(w), None, s, p)
    y, st = Conv(dev).fwd(x, w, s, p, pre=(ps, pt, True), stats=True, bf16=mode)
    # not tighter: the on-load affine is an fma on the GPU and mul+add in torch, and a 1-ulp f32 difference that straddles
    # a bf16 rounding boundary moves that operand by 2^-8 relative (measured 2e-5 .. 1.3e-4 on the layer shapes)
    assert relerr(y, ref) < 5e-4 + OUT_TOL[mode]
    assert relerr(y, F.conv2d(xin, w, None, s, p)) < 2e-2
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)   # statistics come from the f32 accumulators


@pytest.mark.parametrize("mode", MODES)
def test_conv_fwd_bf16_residual_relu(env, mode):
    """epilogue with residual (+ReLU): the residual is read in the tensors' element type"""
    dev, _ = env
    cfg = (2, 6, 8, 64, 64, 3, 1, 1)
    x, w = make(cfg, 40)
    r = torch.randn((2, 64, 6, 8), generator=torch.Generator().manual_seed(41))
    if mode == 2:
        x, r = rbf(x), rbf(r)
    ref = F.relu(F.conv2d(rbf(x), rbf(w), None, 1, 1) + r)
    y, _ = Conv(dev).fwd(x, w, 1, 1, resid=r, relu=1, bf16=mode)
    assert relerr(y, ref) < 1e-4 + OUT_TOL[mode]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cfg", BF_SMALL + [(40, 5, 6, 64, 64, 3, 1, 1)] + BF_REAL)
def test_conv_wgrad_bf16_mode(env, cfg, mode):
    dev, _ = env
    N, H, W, C, K, k, s, p = cfg
    x, w = make(cfg, 22)
    w = w.requires_grad_(True)
    y = F.conv2d(rbf(x), w, None, s, p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(23))
    y.backward(rbf(dy))
    dw = Conv(dev).wgrad(rbf(x) if mode == 2 else x, rbf(dy) if mode == 2 else dy, k, s, p, bf16=mode)
    assert relerr(dw, w.grad) < 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_conv_wgrad_bf16_bn_relu_on_load(env, mode):
    """conv2's weight gradient reads y1 with bn1 + ReLU applied on load (f32), then rounds the operand"""
    dev, _ = env
    cfg = (3, 6, 8, 64, 128, 3, 1, 1)
    x, w = make(cfg, 42)
    g = torch.Generator().manual_seed(43)
    ps, pt = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    if mode == 2:
        x = rbf(x)
    xin = F.relu(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
    w = w.requires_grad_(True)
    y = F.conv2d(rbf(xin), w, None, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(rbf(dy))
    dw = Conv(dev).wgrad(x, rbf(dy) if mode == 2 else dy, 3, 1, 1, pre=(ps, pt, True), bf16=mode)
    assert relerr(dw, w.grad) < 5e-4   # fma vs mul+add before a rounding boundary, see test_conv_fwd_bf16_mode


@pytest.mark.parametrize("cfg", BF_SMALL[:3] + [pytest.param((4, 20, 48, 128, 128, 3, 1, 1), marks=gpu), pytest.param((2, 20, 48, 128, 256, 3, 2, 1), marks=gpu)])
@pytest.mark.parametrize("mode", MODES)
def test_conv_dgrad_bf16_mode_and_transposed_weights(env, cfg, mode):
    dev, _ = env
    N, H, W, C, K, k, s, p = cfg
    x, w = make(cfg, 24)
    x.requires_grad_(True)
    y = F.conv2d(x, rbf(w), None, s, p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(25))
    y.backward(rbf(dy))
    r = rbf(torch.randn(x.shape, generator=torch.Generator().manual_seed(27)))
    dx = Conv(dev).dgrad(rbf(dy) if mode == 2 else dy, w, H, W, s, p, resid=r, bf16=mode, transposed=True)
    assert relerr(dx, x.grad + r) < 1e-4 + OUT_TOL[mode]
    # the transposed-weight route in exact f32 as well
    x.grad = None
    F.conv2d(x, w, None, s, p).backward(dy)
    assert relerr(Conv(dev).dgrad(dy, w, H, W, s, p, bf16=0, transposed=True), x.grad) < 1e-5


@pytest.mark.parametrize("cfg", [(2, 3, 4, 64, 64), (1, 5, 12, 128, 64)] + [pytest.param((4, 5, 12, 640, 256), marks=gpu), pytest.param((2, 20, 48, 128, 64), marks=gpu)])
@pytest.mark.parametrize("mode", MODES)
def test_deconv_bf16_mode(env, cfg, mode):
    dev, _ = env
    N, H, W, C, K = cfg
    g = torch.Generator().manual_seed(26)
    x = torch.randn((N, C, H, W), generator=g)
    if mode == 2:
        x = rbf(x)
    w = (torch.randn((C, K, 3, 3), generator=g) * (2.0 / (C * 2.25)) ** 0.5).requires_grad_(True)
    b = torch.randn(K, generator=g)
    ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xn = (x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1)).requires_grad_(True)
    wr = rbf(w.detach()).requires_grad_(True)
    u = F.conv_transpose2d(rbf(xn), wr, b, 2, 1, 1)
    ref = F.relu(u)
    y, st, bwd = Conv(dev).deconv_all(x, w.detach(), b, (ps, pt), relu=1, bf16=mode)
    assert relerr(y, ref) < 5e-4 + OUT_TOL[mode]          # see test_conv_fwd_bf16_mode
    dy = rbf(torch.randn(u.shape, generator=g))
    # reference backward with the executor's rounding points: dy rounded; dx = gather over rounded dy with rounded w;
    # dw = rounded bn(x) x rounded dy
    xr = rbf(xn.detach()).requires_grad_(True)
    F.conv_transpose2d(xr, wr, None, 2, 1, 1).backward(rbf(dy))
    dx, dw = bwd(dy)
    assert relerr(dx, xr.grad) < 1e-4 + OUT_TOL[mode] and relerr(dw, wr.grad) < 5e-4


# ---- halo-staged 3x3 / stride-1 kernel (conv_halo.hip): bf16 tensors + bf16 weight copies, 128-row tiles ------------------
HALO_SMALL = [(1, 5, 12, 64, 64), (2, 6, 7, 128, 64), (1, 9, 8, 64, 128), (3, 4, 5, 64, 64)]
HALO_REAL = [pytest.param(c, marks=gpu) for c in [(8, 40, 96, 64, 64), (4, 20, 48, 128, 128), (8, 10, 24, 256, 256), (16, 5, 12, 512, 512), (3, 48, 48, 64, 64)]]


@pytest.fixture
def force_cfg(lbc_config):
    yield lambda c: lbc_config("LBC_FORCE_CFG", c)


@pytest.mark.parametrize("cfg", HALO_SMALL + HALO_REAL)
def test_conv3x3_halo_fwd(env, cfg, force_cfg):
    """forward with BatchNorm+ReLU on load, statistics partials, odd widths and tiles straddling images"""
    dev, _ = env
    N, H, W, C, K = cfg
    force_cfg(1 if K % 128 == 0 else 0)
    x, w = make((N, H, W, C, K, 3, 1, 1), 50)
    x = rbf(x)
    g = torch.Generator().manual_seed(51)
    ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xin = F.relu(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
    ref = F.conv2d(rbf(xin), rbf(w), None, 1, 1)
    y, st = Conv(dev).fwd(x, w, 1, 1, pre=(ps, pt, True), stats=True, bf16=3)
    assert relerr(y, ref) < 5e-4 + OUT_TOL[2]
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    # without the prologue, with a residual
    r = rbf(torch.randn(ref.shape, generator=g))
    y2, _ = Conv(dev).fwd(x, w, 1, 1, resid=r, relu=1, bf16=3)
    assert relerr(y2, F.relu(F.conv2d(x, rbf(w), None, 1, 1) + r)) < 1e-4 + OUT_TOL[2]


@pytest.mark.parametrize("cfg", HALO_SMALL[:3] + HALO_REAL[:3])
def test_conv3x3_halo_dgrad(env, cfg, force_cfg):
    dev, _ = env
    N, H, W, C, K = cfg
    force_cfg(1 if C % 128 == 0 else 0)     # the input gradient's output channels are the convolution's input channels
    x, w = make((N, H, W, C, K, 3, 1, 1), 52)
    x.requires_grad_(True)
    y = F.conv2d(x, rbf(w), None, 1, 1)
    dy = rbf(torch.randn(y.shape, generator=torch.Generator().manual_seed(53)))
    y.backward(dy)
    r = rbf(torch.randn(x.shape, generator=torch.Generator().manual_seed(54)))
    dx = Conv(dev).dgrad(dy, w, H, W, 1, 1, resid=r, bf16=3, transposed=True)
    assert relerr(dx, x.grad + r) < 1e-4 + OUT_TOL[2]


@pytest.mark.parametrize("cfg", [(1, 40, 48, 64, 64), pytest.param((8, 40, 96, 64, 64), marks=gpu), pytest.param((4, 20, 48, 128, 128), marks=gpu)])
def test_conv3x3_halo_persistent_workgroups(env, cfg, force_cfg, lbc_config):
    """fewer workgroups than tiles: every workgroup walks several tiles with the next tile's halo prefetched"""
    dev, _ = env
    N, H, W, C, K = cfg
    force_cfg(1 if K % 128 == 0 else 0)
    lbc_config("LBC_HALO_BLOCKS", 8)
    if True:
        x, w = make((N, H, W, C, K, 3, 1, 1), 55)
        x = rbf(x)
        g = torch.Generator().manual_seed(56)
        ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        xin = F.relu(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
        ref = F.conv2d(rbf(xin), rbf(w), None, 1, 1)
        y, st = Conv(dev).fwd(x, w, 1, 1, pre=(ps, pt, True), stats=True, bf16=3)
        assert relerr(y, ref) < 5e-4 + OUT_TOL[2]
        assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
        x.requires_grad_(True)
        yy = F.conv2d(x, rbf(w), None, 1, 1)
        dy = rbf(torch.randn(yy.shape, generator=g))
        yy.backward(dy)
        if C == K:
            dx = Conv(dev).dgrad(dy, w, H, W, 1, 1, bf16=3, transposed=True)
            assert relerr(dx, x.grad) < 1e-4 + OUT_TOL[2]


# ---- tap-fused 3x3 / stride-1 weight gradient with transpose reads (conv_wgrad_tr.hip): bf16 tensors, W % 8 == 0 ----------
WTR_SMALL = [(2, 5, 12, 64, 64), (1, 3, 9, 64, 64), (1, 4, 16, 64, 64), (2, 3, 24, 64, 128), (1, 9, 16, 128, 64), (3, 2, 16, 64, 64), (4, 16, 16, 64, 64)]   # the last: 16 chunks, ring wrap, 2 splits
WTR_REAL = [pytest.param(c, marks=gpu) for c in [(16, 5, 12, 512, 512), (8, 40, 96, 64, 64), (4, 20, 48, 128, 128), (8, 10, 24, 256, 256), (3, 48, 48, 64, 64), (2, 20, 48, 64, 128)]]


@pytest.mark.parametrize("cfg", WTR_SMALL + WTR_REAL)
def test_conv_wgrad_tap_fused(env, cfg):
    """plain and with the producer's BatchNorm+ReLU applied to x on load; image borders, several images per split"""
    dev, _ = env
    N, H, W, C, K = cfg
    x, w = make((N, H, W, C, K, 3, 1, 1), 60)
    x = rbf(x)
    g = torch.Generator().manual_seed(61)
    dy = rbf(torch.randn((N, K, H, W), generator=g))
    w1 = w.clone().requires_grad_(True)
    F.conv2d(x, w1, None, 1, 1).backward(dy)
    dw = Conv(dev).wgrad(x, dy, 3, 1, 1, bf16=2)
    assert relerr(dw, w1.grad) < 1e-4
    ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xin = F.relu(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
    w2 = w.clone().requires_grad_(True)
    F.conv2d(rbf(xin), w2, None, 1, 1).backward(dy)
    dw2 = Conv(dev).wgrad(x, dy, 3, 1, 1, pre=(ps, pt, True), bf16=2)
    assert relerr(dw2, w2.grad) < 5e-4   # fma vs mul+add before a rounding boundary, see test_conv_fwd_bf16_mode


# ---- tap-fused 3x3 / STRIDE-2 weight gradient (conv_wgrad_tr2.hip): the first convolution of layers 2-4 and the decoder's transposed convolutions ----------
WTR2_SMALL = [(2, 8, 24, 64, 128), (1, 10, 48, 128, 128), (3, 6, 32, 64, 256), (2, 24, 24, 64, 128), (1, 4, 96, 64, 128), (5, 2, 24, 128, 128), (1, 34, 40, 64, 128)]
WTR2_REAL = [pytest.param(c, marks=gpu) for c in [(32, 40, 96, 64, 128), (64, 20, 48, 128, 256), (256, 10, 24, 256, 512), (7, 40, 96, 64, 128)]]


@pytest.mark.parametrize("cfg", WTR2_SMALL + WTR2_REAL)
def test_conv_wgrad_stride2_tap_fused(env, cfg, lbc_config):
    """image borders (top / left taps), several images per split, output rows that end inside a 16-pixel group, ring wrap; bit-compared with
    nothing (its summation order is its own): against torch on bf16-rounded operands, and against the generic kernel (LBC_WGRAD_TR2_MIN_WGS huge = never)"""
    dev, _ = env
    lbc_config("LBC_WGRAD_TR2_MIN_WGS", 1)       # (the kernel is selected from ~192 workgroups of work; here at any size)
    N, H, W, C, K = cfg
    x, w = make((N, H, W, C, K, 3, 2, 1), 70)
    x = rbf(x)
    g = torch.Generator().manual_seed(71)
    dy = rbf(torch.randn((N, K, H // 2, W // 2), generator=g))
    w1 = w.clone().requires_grad_(True)
    F.conv2d(x, w1, None, 2, 1).backward(dy)
    dw = Conv(dev).wgrad(x, dy, 3, 2, 1, bf16=2)
    assert relerr(dw, w1.grad) < 1e-4
    lbc_config("LBC_WGRAD_TR2_MIN_WGS", 1 << 40)
    assert relerr(dw, Conv(dev).wgrad(x, dy, 3, 2, 1, bf16=2)) < 1e-5


@pytest.mark.parametrize("cfg", [(2, 5, 12, 128, 64), (1, 7, 16, 256, 128), (3, 4, 48, 128, 64)] +
                         [pytest.param(c, marks=gpu) for c in [(64, 5, 12, 640, 256), (64, 10, 24, 256, 128), (64, 20, 48, 128, 64), (256, 20, 48, 128, 64)]])
def test_deconv_wgrad_stride2_tap_fused(env, cfg, lbc_config):
    """ConvTranspose2d weight gradient on bf16 tensors with the preceding BatchNorm applied to x on load: the same kernel with the roles
    of the two tensors swapped (P = bn(x) on the low-resolution lattice, Q = dY on the high-resolution one)"""
    dev, _ = env
    lbc_config("LBC_WGRAD_TR2_MIN_WGS", 1)
    N, H, W, C, K = cfg
    g = torch.Generator().manual_seed(72)
    x = rbf(torch.randn((N, C, H, W), generator=g))
    w = (torch.randn((C, K, 3, 3), generator=g) * (2.0 / (C * 2.25)) ** 0.5)
    b = torch.randn(K, generator=g)
    ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    _, _, bwd = Conv(dev).deconv_all(x, w, b, (ps, pt), relu=0, bf16=2)
    dy = rbf(torch.randn((N, K, 2 * H, 2 * W), generator=g))
    w1 = w.clone().requires_grad_(True)
    xn = rbf(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
    F.conv_transpose2d(xn, w1, None, 2, 1, 1).backward(dy)
    _, dw = bwd(dy)
    assert relerr(dw, w1.grad) < 5e-4      # (fma vs mul + add in front of the operand's bf16 rounding)
    lbc_config("LBC_WGRAD_TR2_MIN_WGS", 1 << 40)
    _, dw0 = bwd(dy)
    assert relerr(dw, dw0) < 1e-5


@pytest.mark.parametrize("cfg", [(3, 2, 5, 12, 64, 64), (5, 1, 4, 16, 64, 128), (12, 4, 16, 16, 64, 64), (2, 3, 2, 16, 128, 64)] +
                         [pytest.param(c, marks=gpu) for c in [(5, 16, 5, 12, 512, 512), (6, 8, 40, 96, 64, 64), (7, 4, 20, 48, 128, 128), (11, 8, 10, 24, 256, 256)]])
def test_conv_wgrad_group(env, cfg):
    """n same-shaped convolutions in one launch (a ResNet stage's): every member's gradient, plain and with BatchNorm+ReLU on load;
    member by member the same numbers as the single launch up to the summation order of the splits"""
    dev, _ = env
    n, N, H, W, C, K = cfg
    g = torch.Generator().manual_seed(67)
    xs = [rbf(torch.randn((N, C, H, W), generator=g)) for _ in range(n)]
    dys = [rbf(torch.randn((N, K, H, W), generator=g)) for _ in range(n)]
    want = []
    for x, dy in zip(xs, dys):
        w1 = torch.zeros((K, C, 3, 3), requires_grad=True)
        F.conv2d(x, w1, None, 1, 1).backward(dy)
        want.append(w1.grad)
    got = Conv(dev).wgrad_group(xs, dys)
    for i in range(n):
        assert relerr(got[i], want[i]) < 1e-4, i
    assert relerr(got[n - 1], Conv(dev).wgrad(xs[n - 1], dys[n - 1], 3, 1, 1, bf16=2)) < 1e-5
    pres = [(torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)) for _ in range(n)]
    got = Conv(dev).wgrad_group(xs, dys, pres=pres)
    for i in (0, n - 1):
        xin = F.relu(xs[i] * pres[i][0].view(1, -1, 1, 1) + pres[i][1].view(1, -1, 1, 1))
        w2 = torch.zeros((K, C, 3, 3), requires_grad=True)
        F.conv2d(rbf(xin), w2, None, 1, 1).backward(dys[i])
        assert relerr(got[i], w2.grad) < 5e-4, i


@pytest.mark.parametrize("case", [(2, 10, 18, 64, 128, 1), (1, 12, 14, 128, 256, 2), (3, 8, 10, 64, 256, 0), (2, 6, 34, 128, 128, 3), (1, 4, 6, 192, 128, 1),
                                  (2, 10, 18, 64, 128, 5), (2, 6, 34, 128, 128, 6), (1, 4, 6, 192, 128, 6)] +
                         [pytest.param(c, marks=gpu) for c in [(32, 40, 96, 64, 128, -1), (64, 20, 48, 128, 256, -1), (256, 10, 24, 256, 512, -1)]])
def test_conv_glds_stride2_transposed_phases(env, case, lbc_config):
    """input gradient of a stride-2 3x3 convolution on the LDS-DMA kernel: the four output-parity phases in one grid (1 / 2 / 2 / 4
    taps), lattice rows past the border, several images per tile; compared with autograd on the bf16-rounded operands and with the
    register-staged kernel on the same launch"""
    dev, _ = env
    N, H, W, C, K, cfgid = case            # conv C -> K over H x W (even), stride 2: dy is [N, K, H/2, W/2], dx has C channels
    if cfgid >= 0:
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
        lbc_config("LBC_GEMM256_CFG", cfgid)
    x, w = make((N, H, W, C, K, 3, 2, 1), 390 + C + K)
    xg = rbf(x).requires_grad_(True)
    yy = F.conv2d(xg, rbf(w), None, 2, 1)
    g = torch.Generator().manual_seed(391)
    dy = rbf(torch.randn(yy.shape, generator=g))
    yy.backward(dy)
    dx = Conv(dev).dgrad(dy, w, H, W, 2, 1, bf16=3, transposed=True)
    assert relerr(dx, xg.grad) < 1e-4 + OUT_TOL[2]
    lbc_config("LBC_NO_GLDS_PHASED", 1)
    dx2 = Conv(dev).dgrad(dy, w, H, W, 2, 1, bf16=3, transposed=True)
    assert relerr(dx, dx2) < 2.0 ** -7


PHASED_SMALL = [(2, 10, 18, 64, 128), (1, 12, 14, 128, 256), (3, 8, 10, 64, 256), (2, 6, 34, 128, 128), (1, 4, 6, 192, 128), (2, 8, 104, 64, 64), (5, 14, 22, 64, 64)]
PHASED_REAL = [pytest.param(c, marks=gpu) for c in [(32, 40, 96, 64, 128), (64, 20, 48, 128, 256), (256, 10, 24, 256, 512), (256, 40, 96, 64, 128), (32, 20, 48, 128, 256)]]


@pytest.mark.parametrize("case", PHASED_SMALL + PHASED_REAL)
def test_conv_hdmap_phased_transposed(env, case, lbc_config):
    """Round 5: the stride-2 transposed launches on the persistent halo-staged kernel (conv_hdmap_k<.., MODE 2>): a 2 x 2-neighbourhood halo
    per 64-channel slab, nine taps feeding four accumulator sets (one per output-parity phase), all four phases of a lattice position
    written by one tile.  (a) input gradient of a stride-2 3x3 convolution against autograd on the bf16-rounded operands and against the
    generic kernel; one / two workgroups for the whole launch (several tiles per workgroup) bit-identical; (b) ConvTranspose2d forward
    with bias + ReLU + statistics (the decoder's form after its bn_apply pass) against torch."""
    dev, _ = env
    from learningbycheating_amd import _lib
    N, H, W, C, K = case            # conv C -> K over H x W (even), stride 2: dy is [N, K, H/2, W/2], dx has C channels
    lib = _lib.get()
    small = N * H * W < 100000
    if small:
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
        lbc_config("LBC_HDMA_CFG", 4)
    x, w = make((N, H, W, C, K, 3, 2, 1), 590 + C + K)
    xg = rbf(x).requires_grad_(True)
    yy = F.conv2d(xg, rbf(w), None, 2, 1)
    g = torch.Generator().manual_seed(591)
    dy = rbf(torch.randn(yy.shape, generator=g))
    yy.backward(dy)
    rows = ctypes.c_int(0)
    dx = Conv(dev).dgrad(dy, w, H, W, 2, 1, bf16=3, transposed=True)
    assert relerr(dx, xg.grad) < 1e-4 + OUT_TOL[2]
    for wgs in (1, 2):
        lbc_config("LBC_HDMA_PERSIST_WGS", wgs)
        assert torch.equal(Conv(dev).dgrad(dy, w, H, W, 2, 1, bf16=3, transposed=True), dx), wgs
    lbc_config("LBC_HDMA_PERSIST_WGS", -1)
    lbc_config("LBC_NO_GLDS_PHASED", 1)
    dx2 = Conv(dev).dgrad(dy, w, H, W, 2, 1, bf16=3, transposed=True)
    assert relerr(dx, dx2) < 2.0 ** -7
    lbc_config("LBC_NO_GLDS_PHASED", -1)
    # (b) ConvTranspose2d(K -> C) forward over the low-resolution map [N, K, H/2, W/2] -> [N, C, H, W]: bias + ReLU + statistics
    LH, LW = H // 2, W // 2
    xt = rbf(torch.randn((N, K, LH, LW), generator=g))
    wt = torch.randn((K, C, 3, 3), generator=g) * (2.0 / (K * 2.25)) ** 0.5
    b = torch.randn(C, generator=g)
    ref = F.relu(F.conv_transpose2d(xt, rbf(wt), b, 2, 1, 1))
    dd = _lib.ConvDesc(N, LH, LW, K, C, 3, 3, 2, 1, 1, 3, 1)
    xh = xt.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    wh = wt.permute(0, 2, 3, 1).contiguous().to(dev)                       # [K][kh][kw][C] = [Cin_T][T][Cout_T]
    wfwd = Conv(dev).transpose(wh.view(K, 9, C), K, 9, C).to(torch.bfloat16)
    bd = b.to(dev)
    _lib.check(lib.lbc_deconv3x3s2_fwd(ctypes.byref(dd), None, None, None, None, None, 0, None, None, ctypes.byref(rows), None))
    assert rows.value == 4 * -(-(N * LH * LW) // 128), (rows.value, N * LH * LW)       # (the four-wave persistent shape: 128 lattice rows per tile, four phases)
    st = torch.zeros((rows.value, 2, C), device=dev)
    from tests.helpers import guarded, check_guard
    buf, y = guarded((N, H, W, C), dev, dtype=torch.bfloat16)
    _lib.check(lib.lbc_deconv3x3s2_fwd(ctypes.byref(dd), _lib.ptr(xh), _lib.ptr(wfwd), _lib.ptr(bd), None, None, 0, _lib.ptr(y), _lib.ptr(st),
                                       ctypes.byref(rows), _lib.stream_for(xh)))
    check_guard(buf, y.numel())
    got = y.permute(0, 3, 1, 2).float().cpu()
    assert relerr(got, ref) < 1e-4 + OUT_TOL[2]
    assert torch.allclose(st.cpu()[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-3, atol=2e-2)
    assert torch.allclose(st.cpu()[:, 1].sum(0), (ref * ref).sum((0, 2, 3)), rtol=1e-3, atol=2e-2)


def test_conv_launch_policy_at_the_per_gpu_batches(env, lbc_config):
    """which tile shape a 3x3 / stride-1 launch of the ResNet-34 layers (resnet.py:164) gets at the per-GPU batches of the 1 / 2 / 4 / 8
    GPU runs -- host logic only, read off the statistics-row count of a query call (rows = M / tile rows).  Eight-wave 256 x 128 tiles
    where they fill the CUs; launches that would be fewer than 160 of them take the four-wave 128 x 64 shape where its 184-row halo holds
    the image rows (layers 3 / 4; measured in profiles/r04_run16_small_tiles_at_120.log); LBC_HDMA_SMALL_BELOW=0 switches that off."""
    from learningbycheating_amd import _lib
    lib = _lib.get()

    def rows(N, H, W, C, K):
        r = ctypes.c_int(0)
        d = _lib.ConvDesc(N, H, W, C, K, 3, 3, 1, 1, 0, 3, 0)
        _lib.check(lib.lbc_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None, 0, None, None, ctypes.byref(r), None))
        return r.value, N * H * W

    L2, L3, L4 = (20, 48, 128, 128), (10, 24, 256, 256), (5, 12, 512, 512)
    expect = {(256, L2): 256, (256, L3): 256, (256, L4): 256,      # 960 / 480 / 240 eight-wave tiles
              (128, L2): 256, (128, L3): 256, (128, L4): 128,      # layer 4: 120 eight-wave tiles -> 480 four-wave tiles
              (64, L2): 256, (64, L3): 128, (64, L4): 128,         # layer 3: 120 -> 480; layer 4: 60 eight-wave tiles are below the fill threshold anyway
              (32, L2): 256, (32, L3): 128, (32, L4): 128}         # layer 2 (120 tiles): its 48-pixel rows need a 226-row halo, the four-wave shape holds 184
    for (N, shape), bm in expect.items():
        r, M = rows(N, *shape)
        assert r == -(-M // bm), (N, shape, r, M, bm)
    lbc_config("LBC_HDMA_SMALL_BELOW", 0)
    assert rows(128, *L4)[0] == -(-128 * 60 // 256) and rows(64, *L3)[0] == -(-64 * 240 // 256)
    assert rows(64, *L4)[0] == -(-64 * 60 // 128)                  # (below the eight-wave fill threshold: unchanged)


# ---- halo-staged LDS-DMA convolution (conv_hdma.hip): 3x3 stride 1, bf16 tensors + bf16 weight copies -------------------------
HDMA_BM = {1: 256, 2: 128, 3: 256, 4: 128}      # tile rows of LBC_HDMA_CFG 1: 256x128, 2: 128x256, 3: 256x64 (persistent, C = K = 64), 4: 128x64 (four waves)
HDMA_SMALL = [(2, 9, 17, 64, 256, 2), (1, 12, 13, 128, 256, 2), (3, 7, 9, 64, 128, 1), (2, 16, 48, 128, 128, 1), (5, 9, 13, 128, 128, 1),
              (1, 20, 24, 64, 512, 1), (1, 3, 30, 192, 256, 2), (2, 30, 12, 64, 256, 2), (3, 20, 24, 128, 256, 1), (2, 13, 30, 64, 512, 2), (3, 10, 24, 128, 128, 4), (2, 5, 12, 192, 256, 4), (1, 9, 27, 64, 192, 4),
              (2, 6, 12, 512, 128, 4),
              (2, 9, 17, 64, 64, 3), (5, 12, 40, 64, 64, 3), (1, 7, 96, 64, 64, 3)]       # the last three: several tiles per persistent workgroup needs LBC_HALO_BLOCKS-like forcing on the GPU only
HDMA_REAL = [pytest.param(c, marks=gpu) for c in [(32, 20, 48, 128, 128, -1), (64, 10, 24, 256, 256, -1), (256, 5, 12, 512, 512, -1),
                                                   (64, 24, 24, 128, 128, -1), (16, 12, 12, 256, 256, 2), (8, 6, 6, 512, 512, 1),
                                                   (32, 40, 96, 64, 64, -1), (40, 48, 48, 64, 64, -1),
                                                   (32, 5, 12, 512, 512, 4), (32, 10, 24, 256, 256, 4)]]      # (the last two: layer 4 / 3 at 32 images, split-K)


@pytest.mark.parametrize("case", HDMA_SMALL + HDMA_REAL)
def test_conv_hdma_fwd_dgrad(env, case, lbc_config):
    """forward (statistics; residual + ReLU) and input gradient (flipped taps, identity gradient added) of the halo-staged kernel
    against f32 convolutions of the bf16-rounded operands: ragged M tails, image borders and several images inside a tile, one
    to eight channel slabs, one or two column tiles, halo rows before / after the tensor"""
    dev, _ = env
    from learningbycheating_amd import _lib
    N, H, W, C, K, cfgid = case
    if cfgid >= 0:
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
        lbc_config("LBC_HDMA_CFG", cfgid)
    if cfgid == 3:
        lbc_config("LBC_HALO_BLOCKS", 2)       # two persistent workgroups: several tiles each (the halo double buffer)
    if cfgid == 4:
        lbc_config("LBC_HDMAP_SPLIT", 0)       # the plain form first (what the bit-for-bit comparisons below are about); the K splits at the end
    x, w = make((N, H, W, C, K, 3, 1, 1), 290 + C + K)
    x = rbf(x)
    ref = F.conv2d(x, rbf(w), None, 1, 1)
    rows = ctypes.c_int(0)
    d = _lib.ConvDesc(N, H, W, C, K, 3, 3, 1, 1, 0, 3, 0)
    _lib.check(_lib.get().lbc_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None, 0, None, None, ctypes.byref(rows), None))
    M = N * H * W
    c64p_rows = lambda bm, cap: -(-(-(-M // bm)) // -(-(-(-M // bm)) // cap))     # one statistics row per persistent workgroup
    if cfgid == 3:      # the 64-channel persistent kernel, LBC_HALO_BLOCKS = 2 workgroups (128-pixel tiles by default, LBC_C64P_BM=256: 256)
        assert rows.value in (c64p_rows(128, 2), c64p_rows(256, 2)), (rows.value, M)
    elif cfgid < 0 and C == 64 and K == 64 and rows.value in (c64p_rows(128, 512), c64p_rows(256, 256)) and rows.value not in (-(-M // 128), -(-M // 256)):
        pass            # (the same kernel selected by the policy at a reference-sized launch)
    else:
        assert rows.value in ([-(-M // HDMA_BM[cfgid])] if cfgid >= 0 else [-(-M // b) for b in (128, 256)]), (rows.value, M)
    y, st = Conv(dev).fwd(x, w, 1, 1, stats=True, bf16=3)
    assert relerr(y, ref) < 1e-4 + OUT_TOL[2]
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    assert torch.allclose(st[:, 1].sum(0), (ref * ref).sum((0, 2, 3)), rtol=1e-3)
    g = torch.Generator().manual_seed(291)
    r = rbf(torch.randn(ref.shape, generator=g))
    y2, _ = Conv(dev).fwd(x, w, 1, 1, resid=r, relu=1, bf16=3)
    assert relerr(y2, F.relu(ref + r)) < 1e-4 + OUT_TOL[2]
    xg = x.clone().requires_grad_(True)
    yy = F.conv2d(xg, rbf(w), None, 1, 1)
    dy = rbf(torch.randn(yy.shape, generator=g))
    yy.backward(dy)
    if K % 64 == 0 and (C % 128 == 0 or C == K == 64 or (cfgid == 4 and C % 64 == 0)):        # the input gradient's output channels are C: needs a column tile of 128 / 256 (64 for the four-wave shape or the 64-channel kernel)
        rr = rbf(torch.randn(x.shape, generator=g))
        dx = Conv(dev).dgrad(dy, w, H, W, 1, 1, resid=rr, bf16=3, transposed=True)
        assert relerr(dx, xg.grad + rr) < 1e-4 + OUT_TOL[2]
    if C == 64 and K == 64 and cfgid in (3, -1):
        # the 64-channel persistent kernel transforms its staged halo in place (conv_c64p_k<0, 0, true>): against the reference, and against
        # the register-staged kernel it replaces (LBC_NO_C64P_PRE=1 -> conv_halo.hip); same rounding points
        ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        xin = rbf(F.relu(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1)))
        refp = F.conv2d(xin, rbf(w), None, 1, 1)
        yp, stp = Conv(dev).fwd(x, w, 1, 1, pre=(ps, pt, True), stats=True, bf16=3)
        # (one statistics row per persistent workgroup: the kernel under test ran)
        assert stp.shape[0] in ((c64p_rows(128, 2), c64p_rows(256, 2)) if cfgid == 3 else (c64p_rows(128, 512), c64p_rows(256, 256)))
        assert relerr(yp, refp) < 5e-4 + OUT_TOL[2]
        assert torch.allclose(stp[:, 0].sum(0), refp.sum((0, 2, 3)), rtol=2e-3, atol=2e-2)
        lbc_config("LBC_NO_C64P_PRE", 1)
        yq, _ = Conv(dev).fwd(x, w, 1, 1, pre=(ps, pt, True), bf16=3)
        assert relerr(yp, yq) < 2.0 ** -7
        lbc_config("LBC_NO_C64P_PRE", 0)
    # The PERSISTENT kernel (conv_hdmap.hpp) with one / two workgroups for the whole launch: a workgroup walks several tiles (halo and
    # weight prefetch across the tile boundary, wave-private copy-out, stores in flight under the next tile) -- same MFMAs in the same
    # order, same order of the statistics sums -> bit-identical outputs and statistics rows
    if cfgid in (1, 2, 4):
        have_dx = K % 64 == 0 and (C % 128 == 0 or (cfgid == 4 and C % 64 == 0))
        for opt, val in (("LBC_HDMA_PERSIST_WGS", 1), ("LBC_HDMA_PERSIST_WGS", 2)):
            lbc_config(opt, val)
            yb, stb = Conv(dev).fwd(x, w, 1, 1, stats=True, bf16=3)
            assert torch.equal(yb, y) and torch.equal(stb, st), (opt, val)
            y2b, _ = Conv(dev).fwd(x, w, 1, 1, resid=r, relu=1, bf16=3)
            assert torch.equal(y2b, y2), (opt, val)
            if have_dx:
                assert torch.equal(Conv(dev).dgrad(dy, w, H, W, 1, 1, resid=rr, bf16=3, transposed=True), dx), (opt, val)
            lbc_config(opt, -1)
            lbc_config("LBC_HDMA_PERSIST_WGS", -1)
    # Split-K of the four-wave shape (lbc_conv_desc.split_workspace; the deep layers at the per-GPU batches of the 8-GPU run): every range
    # count that divides the channel slabs, against the reference and against the unsplit launch -- same products, the f32 sums
    # regrouped (per range, then over the ranges): outputs within one bf16 rounding, statistics rows (same row count) within f32 noise
    if cfgid == 4:
        for ns in [n for n in (2, 3, 4, 8) if (C // 64) % n == 0]:
            lbc_config("LBC_HDMAP_SPLIT", ns)
            cs = Conv(dev, split_floats=ns * M * max(C, K))
            ys, sts = cs.fwd(x, w, 1, 1, stats=True, bf16=3)
            assert torch.isfinite(cs.split_ws[:ns * M * K]).all() and torch.isnan(cs.split_ws[ns * M * K:]).all(), "the split launch did not run (or wrote outside its partial tiles)"
            assert relerr(ys, ref) < 1e-4 + OUT_TOL[2] and relerr(ys, y) < 2.0 ** -7
            assert sts.shape == st.shape and torch.allclose(sts, st, rtol=1e-4, atol=1e-3), (sts - st).abs().max()
            ys2, _ = cs.fwd(x, w, 1, 1, resid=r, relu=1, bf16=3)
            assert relerr(ys2, F.relu(ref + r)) < 1e-4 + OUT_TOL[2] and relerr(ys2, y2) < 2.0 ** -7
            if K % 64 == 0 and C % 64 == 0 and (K // 64) % ns == 0:
                dxs = cs.dgrad(dy, w, H, W, 1, 1, resid=rr, bf16=3, transposed=True)
                assert relerr(dxs, xg.grad + rr) < 1e-4 + OUT_TOL[2] and relerr(dxs, dx) < 2.0 ** -7
            # a scratch too small for the partial tiles: the launch stays unsplit (bit-identical to it)
            small = Conv(dev, split_floats=ns * M * K - 1)
            yb, stb = small.fwd(x, w, 1, 1, stats=True, bf16=3)
            assert torch.equal(yb, y) and torch.equal(stb, st) and torch.isnan(small.split_ws).all()
            lbc_config("LBC_HDMAP_SPLIT", 0)
        # The in-workgroup K split (round 5; conv_hdmap_k<.., KG = 2>: the policy's choice for launches of at most one tile per CU with an
        # even number of channel slabs -- every case here that has one): two four-wave instances per workgroup contract half the slabs
        # each, instance 1 hands its accumulators over through LDS.  Same products, the f32 sums regrouped once: every epilogue form
        # against the reference and within one bf16 rounding of the plain launch; statistics rows within f32 noise
        lbc_config("LBC_HDMAP_SPLIT", -1)
        yk, stk = Conv(dev).fwd(x, w, 1, 1, stats=True, bf16=3)
        assert relerr(yk, ref) < 1e-4 + OUT_TOL[2] and relerr(yk, y) < 2.0 ** -7
        assert stk.shape == st.shape and torch.allclose(stk, st, rtol=1e-4, atol=1e-3), (stk - st).abs().max()
        if (C // 64) % 2 == 0:
            assert not torch.equal(stk, st), "the K-split launch did not run (its f32 sums are grouped differently)"
        yk2, _ = Conv(dev).fwd(x, w, 1, 1, resid=r, relu=1, bf16=3)
        assert relerr(yk2, F.relu(ref + r)) < 1e-4 + OUT_TOL[2] and relerr(yk2, y2) < 2.0 ** -7
        if K % 64 == 0 and C % 64 == 0:
            dxk = Conv(dev).dgrad(dy, w, H, W, 1, 1, resid=rr, bf16=3, transposed=True)
            assert relerr(dxk, xg.grad + rr) < 1e-4 + OUT_TOL[2] and relerr(dxk, dx) < 2.0 ** -7
    # A/B: the per-tap LDS-DMA kernel on the same launch gives the same result up to summation order
    lbc_config("LBC_NO_HDMA", 1)
    y3, _ = Conv(dev).fwd(x, w, 1, 1, bf16=3)
    assert relerr(y, y3) < 2.0 ** -7


@pytest.mark.parametrize("case", [(2, 10, 18, 64, 128, 3, 1), (1, 12, 14, 128, 256, 3, 2), (3, 8, 10, 64, 128, 1, 3), (2, 6, 34, 128, 128, 3, 3),
                                  (2, 10, 18, 64, 128, 3, 6), (3, 8, 10, 64, 128, 1, 6), (2, 6, 34, 128, 64, 3, 5)] +
                         [pytest.param(c, marks=gpu) for c in [(32, 40, 96, 64, 128, 3, -1), (64, 20, 48, 128, 256, 3, -1), (64, 40, 96, 64, 128, 1, -1), (256, 10, 24, 256, 512, 3, -1)]])
def test_conv_glds_stride2_gather(env, case, lbc_config):
    """stride-2 forward (3x3 pad 1 and the 1x1 downsample) on the LDS-DMA kernel: odd output extents, borders, statistics"""
    dev, _ = env
    from learningbycheating_amd import _lib
    N, H, W, C, K, k, cfgid = case
    if cfgid >= 0:
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
        lbc_config("LBC_GEMM256_CFG", cfgid)
    p = (k - 1) // 2
    x, w = make((N, H, W, C, K, k, 2, p), 190 + C + K)
    x = rbf(x)
    ref = F.conv2d(x, rbf(w), None, 2, p)
    rows = ctypes.c_int(0)
    d = _lib.ConvDesc(N, H, W, C, K, k, k, 2, p, 0, 3, 0)
    _lib.check(_lib.get().lbc_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None, 0, None, None, ctypes.byref(rows), None))
    M = N * ref.shape[2] * ref.shape[3]
    assert rows.value in ([-(-M // GLDS_BM[cfgid])] if cfgid >= 0 else [-(-M // b) for b in (128, 256, 512)]), (rows.value, M)
    y, st = Conv(dev).fwd(x, w, 2, p, stats=True, bf16=3)
    assert relerr(y, ref) < 1e-4 + OUT_TOL[2]
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    lbc_config("LBC_NO_GEMM256", 1)
    y3, _ = Conv(dev).fwd(x, w, 2, p, bf16=3)
    assert relerr(y, y3) < 2.0 ** -7


# ---- randomized small shapes on the emulator (and the GPU): ragged pixel counts, odd widths, images smaller than a tile -----
def _rand_shapes(seed, count, wmin, wmax, wstep=1):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(count):
        n = int(torch.randint(1, 4, (1,), generator=g))
        h = int(torch.randint(1, 9, (1,), generator=g))
        w = int(torch.randint(wmin // wstep, wmax // wstep + 1, (1,), generator=g)) * wstep
        out.append((n, h, w))
    return out


@pytest.mark.parametrize("shape", _rand_shapes(70, 6, 3, 20))
def test_conv3x3_c64_random_shapes(env, shape, force_cfg):
    dev, _ = env
    N, H, W = shape
    force_cfg(0)
    x, w = make((N, H, W, 64, 64, 3, 1, 1), 71 + H * 31 + W)
    x = rbf(x)
    ref = F.conv2d(x, rbf(w), None, 1, 1)
    y, st = Conv(dev).fwd(x, w, 1, 1, stats=True, bf16=3)
    assert relerr(y, ref) < 1e-4 + OUT_TOL[2]
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    x.requires_grad_(True)
    yy = F.conv2d(x, rbf(w), None, 1, 1)
    dy = rbf(torch.randn(yy.shape, generator=torch.Generator().manual_seed(72)))
    yy.backward(dy)
    dx = Conv(dev).dgrad(dy, w, H, W, 1, 1, bf16=3, transposed=True)
    assert relerr(dx, x.grad) < 1e-4 + OUT_TOL[2]


@pytest.mark.parametrize("shape", _rand_shapes(76, 8, 12, 48, 4))
def test_conv_wgrad_stride2_tap_fused_random_shapes(env, shape, lbc_config):
    """the stride-2 tap-fused weight gradient on random (images, output rows, output width % 4 == 0 >= 12): pixel counts that do not fill
    the last 32-pixel chunk, output rows ending anywhere inside a 16-pixel group, ring wrap after a few chunks"""
    dev, _ = env
    lbc_config("LBC_WGRAD_TR2_MIN_WGS", 1)
    N, OH, OW = shape
    H, W = 2 * OH, 2 * OW
    x, w = make((N, H, W, 64, 128, 3, 2, 1), 77 + OH * 19 + OW)
    x = rbf(x)
    dy = rbf(torch.randn((N, 128, OH, OW), generator=torch.Generator().manual_seed(78)))
    w1 = w.clone().requires_grad_(True)
    F.conv2d(x, w1, None, 2, 1).backward(dy)
    assert relerr(Conv(dev).wgrad(x, dy, 3, 2, 1, bf16=2), w1.grad) < 1e-4


@pytest.mark.parametrize("shape", _rand_shapes(73, 6, 8, 24))
def test_conv_wgrad_tap_fused_random_shapes(env, shape):
    """widths with W % 8 == 0, W % 4 == 0 and neither; pixel counts that do not fill the last 64-pixel chunk"""
    dev, _ = env
    N, H, W = shape
    x, w = make((N, H, W, 64, 128, 3, 1, 1), 74 + H * 17 + W)
    x = rbf(x)
    dy = rbf(torch.randn((N, 128, H, W), generator=torch.Generator().manual_seed(75)))
    w1 = w.clone().requires_grad_(True)
    F.conv2d(x, w1, None, 1, 1).backward(dy)
    assert relerr(Conv(dev).wgrad(x, dy, 3, 1, 1, bf16=2), w1.grad) < 1e-4


@pytest.mark.parametrize("cfgid", [0, 1, 2])
@pytest.mark.parametrize("shape", _rand_shapes(76, 3, 3, 14))
def test_conv_generic_bf16_weights_random_shapes(env, shape, cfgid, force_cfg):
    """the prefetch-distance-2 pipeline of the all-bf16 generic kernel on every tile configuration, incl. depth chunks < 2"""
    dev, _ = env
    N, H, W = shape
    force_cfg(cfgid)
    for (C, K, k, s_, p_) in [(64, 128, 1, 1, 0), (128, 128, 3, 1, 1), (64, 128, 3, 2, 1)]:
        if s_ == 2 and (H % 2 or W % 2):
            continue
        x, w = make((N, H, W, C, K, k, s_, p_), 77 + H + W + C)
        x = rbf(x)
        ref = F.conv2d(x, rbf(w), None, s_, p_)
        y, _ = Conv(dev).fwd(x, w, s_, p_, bf16=3)
        assert relerr(y, ref) < 1e-4 + OUT_TOL[2]


# ---- exact-f32 path on the large-batch tile configurations --------------------------------------------------------------
# lbc_igemm_pick() takes 128 x 128 / 128 x 64 tiles once a launch has >= 384 of them, i.e. at the batch sizes bench.py and the
# BASELINE configs run (64 .. 256); test-sized batches fall through to 64 x 64.  LBC_FORCE_CFG pins the policy so the very
# kernels of the f32 headline ("1e-3 waypoint parity") are compared with torch here: forward (gather mode) with the fused
# prologue / epilogue, input gradient (transposed mode) in both weight layouts, the stride-2 phase launches, ragged M tails.
F32_TILE_SMALL = [(2, 9, 8, 64, 128, 3, 1, 1), (1, 10, 12, 64, 128, 3, 2, 1), (3, 5, 7, 128, 128, 3, 1, 1), (1, 8, 12, 64, 128, 1, 2, 0),
                  (1, 4, 5, 32, 64, 3, 1, 1)]
F32_TILE_REAL = [pytest.param(c, marks=gpu) for c in [
    (32, 40, 96, 64, 64, 3, 1, 1), (32, 20, 48, 128, 128, 3, 1, 1), (64, 10, 24, 256, 256, 3, 1, 1), (64, 5, 12, 512, 512, 3, 1, 1),
    (32, 40, 96, 64, 128, 3, 2, 1), (32, 40, 96, 64, 128, 1, 2, 0), (32, 20, 48, 128, 256, 3, 2, 1)]]


@pytest.mark.parametrize("cfgid", [0, 1])
@pytest.mark.parametrize("cfg", F32_TILE_SMALL + F32_TILE_REAL)
def test_conv_f32_large_tile_configs(env, cfg, cfgid, force_cfg):
    dev, _ = env
    N, H, W, C, K, k, s, p = cfg
    force_cfg(cfgid)
    x, w = make(cfg, 80 + cfgid)
    g = torch.Generator().manual_seed(81)
    # forward with BatchNorm+ReLU on load, bias, residual, ReLU and the statistics partials
    ps, pt, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g), torch.randn(K, generator=g)
    xin = F.relu(x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1))
    ref = F.conv2d(xin, w, b, s, p)
    r = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref + r)
    y, st = Conv(dev).fwd(x, w, s, p, bias=b, resid=r, pre=(ps, pt, True), relu=1, stats=True)
    assert relerr(y, ref) < 1e-5
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-4 * ref.abs().sum((0, 2, 3)).max().item())
    assert torch.allclose(st[:, 1].sum(0), (ref * ref).sum((0, 2, 3)), rtol=1e-4)
    # plain forward (no prologue): the path of conv1 / the downsample
    y0, _ = Conv(dev).fwd(x, w, s, p)
    assert relerr(y0, F.conv2d(x, w, None, s, p)) < 1e-5
    # input gradient, transposed mode, with the identity gradient added in the epilogue; both weight layouts
    xg = x.clone().requires_grad_(True)
    yy = F.conv2d(xg, w, None, s, p)
    dy = torch.randn(yy.shape, generator=g)
    yy.backward(dy)
    rr = torch.randn(x.shape, generator=g)
    if C % 64:
        return                                # (no network layer has fewer than 64 gathered channels on the gradient side)
    if not (k == 1 and s == 2):
        assert relerr(Conv(dev).dgrad(dy, w, H, W, s, p), xg.grad) < 1e-5
        if C % 128 == 0 or cfgid == 0:      # the input gradient's output channels are the convolution's input channels
            assert relerr(Conv(dev).dgrad(dy, w, H, W, s, p, resid=rr, transposed=True), xg.grad + rr) < 1e-5
    else:
        assert relerr(Conv(dev).dgrad(dy, w, H, W, s, p, resid=rr), xg.grad + rr) < 1e-5    # 1x1/2: only even pixels are written


@pytest.mark.parametrize("cfgid", [0, 1])
@pytest.mark.parametrize("cfg", [(2, 5, 12, 128, 128), (1, 9, 8, 64, 64)] + [pytest.param((32, 5, 12, 640, 256), marks=gpu), pytest.param((32, 10, 24, 256, 128), marks=gpu),
                                                                              pytest.param((32, 20, 48, 128, 64), marks=gpu)])
def test_deconv_f32_large_tile_configs(env, cfg, cfgid, force_cfg):
    """ConvTranspose2d forward = four output-parity phases of the transposed mode in one launch, on 128-row tiles"""
    dev, _ = env
    N, H, W, C, K = cfg
    if cfgid == 1 and K % 128:
        pytest.skip("128-column tiles need K % 128 == 0")
    force_cfg(cfgid)
    g = torch.Generator().manual_seed(82)
    x = torch.randn((N, C, H, W), generator=g)
    w = (torch.randn((C, K, 3, 3), generator=g) * (2.0 / (C * 2.25)) ** 0.5).requires_grad_(True)
    b = torch.randn(K, generator=g)
    ps, pt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xn = (x * ps.view(1, -1, 1, 1) + pt.view(1, -1, 1, 1)).requires_grad_(True)
    u = F.conv_transpose2d(xn, w, b, 2, 1, 1)
    ref = F.relu(u)
    y, st, bwd = Conv(dev).deconv_all(x, w.detach(), b, (ps, pt), relu=1)
    assert relerr(y, ref) < 1e-5
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    dy = torch.randn(u.shape, generator=g)
    u.backward(dy)
    dx, dw = bwd(dy)
    assert relerr(dx, xn.grad) < 1e-5 and relerr(dw, w.grad) < 2e-5


# ---- 8-wave LDS-DMA convolution (conv_glds.hip): bf16 tensors + bf16 weight copies --------------------------------------------
GLDS_BM = {0: 256, 1: 256, 2: 128, 3: 512, 4: 512, 5: 256, 6: 128}      # tile rows of LBC_GEMM256_CFG 0: 256x256, 1: 256x128, 2: 128x256, 3: 512x128, 4: 512x64; four waves, two workgroups per CU: 5: 256x64, 6: 128x128


def _glds_cases():
    out = []
    for (N, H, W, C, K, k) in [(2, 9, 17, 64, 256, 3), (1, 12, 13, 128, 256, 3), (3, 7, 9, 64, 128, 3), (1, 10, 30, 64, 256, 1), (2, 16, 17, 128, 128, 3),
                               (5, 9, 13, 128, 128, 3)]:
        for cfgid in ((0, 2) if K % 256 == 0 else (1, 3)):
            out.append((N, H, W, C, K, k, cfgid))
    out += [(2, 9, 17, 64, 64, 3, 4), (1, 30, 20, 128, 64, 3, 4)]      # 512 x 64 tiles (second-generation kernel only)
    out += [(2, 9, 17, 64, 64, 3, 5), (1, 30, 20, 128, 64, 3, 5), (3, 7, 9, 64, 128, 3, 6), (2, 16, 17, 128, 128, 3, 6), (1, 10, 30, 64, 256, 1, 6)]      # four-wave shapes
    return out


# (cfg -1: the shape the cost model picks at the default tile-count threshold)
GLDS_REAL = [pytest.param(c, marks=gpu) for c in [(32, 20, 48, 128, 128, 3, -1), (64, 10, 24, 256, 256, 3, -1), (256, 5, 12, 512, 512, 3, -1),
                                                   (64, 24, 24, 128, 128, 3, -1), (4, 20, 48, 128, 256, 3, 0), (5, 10, 24, 256, 512, 3, 2),
                                                   (7, 20, 48, 128, 128, 3, 1), (7, 20, 48, 128, 128, 3, 3)]]


@pytest.mark.parametrize("case", _glds_cases() + GLDS_REAL)
def test_conv_glds_fwd_dgrad(env, case, lbc_config):
    """forward with the epilogue variants (statistics; residual + ReLU) and the input gradient (flipped taps, identity
    gradient added) against f32 convolutions of the bf16-rounded operands, for every tile shape; ragged M tails, image
    borders inside a tile, several images per tile, 1 .. 18 depth steps"""
    dev, _ = env
    from learningbycheating_amd import _lib
    N, H, W, C, K, k, cfgid = case
    lbc_config("LBC_NO_HDMA", 1)           # this test is about conv_glds.hip (3x3 stride-1 launches prefer conv_hdma.hip otherwise)
    if cfgid >= 0:
        lbc_config("LBC_GEMM256_MIN_TILES", 1)
        lbc_config("LBC_GEMM256_CFG", cfgid)
    p = (k - 1) // 2
    x, w = make((N, H, W, C, K, k, 1, p), 90 + C + K)
    x = rbf(x)
    ref = F.conv2d(x, rbf(w), None, 1, p)
    # 1. plain forward + statistics.  The kernel must actually be the one under test: its partial-row count is M / BM
    rows = ctypes.c_int(0)
    d = _lib.ConvDesc(N, H, W, C, K, k, k, 1, p, 0, 3, 0)
    _lib.check(_lib.get().lbc_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None, 0, None, None, ctypes.byref(rows), None))
    M = N * H * W
    assert rows.value in ([-(-M // GLDS_BM[cfgid])] if cfgid >= 0 else [-(-M // b) for b in (128, 256, 512)]), (rows.value, M)
    y, st = Conv(dev).fwd(x, w, 1, p, stats=True, bf16=3)
    assert relerr(y, ref) < 1e-4 + OUT_TOL[2]
    assert torch.allclose(st[:, 0].sum(0), ref.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    assert torch.allclose(st[:, 1].sum(0), (ref * ref).sum((0, 2, 3)), rtol=1e-3)
    # 2. residual + ReLU epilogue
    g = torch.Generator().manual_seed(91)
    r = rbf(torch.randn(ref.shape, generator=g))
    y2, _ = Conv(dev).fwd(x, w, 1, p, resid=r, relu=1, bf16=3)
    assert relerr(y2, F.relu(ref + r)) < 1e-4 + OUT_TOL[2]
    # 3. input gradient (the transposed mode) with the identity gradient in the epilogue
    if C % 128 == 0:
        xg = x.clone().requires_grad_(True)
        yy = F.conv2d(xg, rbf(w), None, 1, p)
        dy = rbf(torch.randn(yy.shape, generator=g))
        yy.backward(dy)
        rr = rbf(torch.randn(x.shape, generator=g))
        dx = Conv(dev).dgrad(dy, w, H, W, 1, p, resid=rr, bf16=3, transposed=True)
        assert relerr(dx, xg.grad + rr) < 1e-4 + OUT_TOL[2]
    # 4. A/B: the generic kernel on the same launch gives the same result up to summation order
    lbc_config("LBC_NO_GEMM256", 1)
    y3, _ = Conv(dev).fwd(x, w, 1, p, bf16=3)
    assert relerr(y, y3) < 2.0 ** -7
