"""Per-kernel parity of the HBM-bound operators against torch CPU, through the single-operator C ABI (include/lbc_hip.h):
BatchNorm2d train / eval forward, running statistics, backward, bn+relu+maxpool forward /
backward (row 3), the waypoint head incl. the SpatialSoftmax corner known-answers of reference common.py:192-201 (row 4),
and the 7x7/2 stem with its fused input pass.  Unmarked cases run the kernel sources under the CPU emulator; gpu-marked
cases run the gfx950 library at the layer shapes of the 160x384 network."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from learningbycheating_amd import _lib
from oracle import lbc_oracle as O
from tests.helpers import guarded, check_guard, nhwc, nchw, relerr

gpu = pytest.mark.gpu
P = _lib.ptr


def _at(bf):
    return torch.bfloat16 if bf else torch.float32


def rbf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _stream(t):
    return _lib.stream_for(t)


# ---------------------------------------------------------------------------------------------------------------
# BatchNorm2d forward: statistics -> finalize (running stats, num_batches_tracked) -> apply (+ residual, + ReLU)
# ---------------------------------------------------------------------------------------------------------------
BN_SHAPES = [(3, 64, 5, 7), (2, 128, 4, 6), (4, 640, 2, 3)]
BN_REAL = [pytest.param(s, marks=gpu) for s in [(32, 64, 40, 96), (32, 128, 20, 48), (64, 256, 10, 24), (64, 512, 5, 12), (32, 640, 5, 12)]]


@pytest.mark.parametrize("bf", [0, 1])
@pytest.mark.parametrize("shape", BN_SHAPES + BN_REAL)
def test_bn_train_forward_and_running_stats(env, shape, bf):
    dev, _ = env
    lib = _lib.get()
    N, C, H, W = shape
    g = torch.Generator().manual_seed(100 + C)
    x = torch.randn(shape, generator=g) * 2 + 0.5
    r = torch.randn(shape, generator=g)
    if bf:
        x, r = rbf(x), rbf(r)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.relu(F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5) + r)
    pixels = N * H * W
    xd, rd = nhwc(x).to(dev).to(_at(bf)), nhwc(r).to(dev).to(_at(bf))
    rows = ctypes.c_int(0)
    _lib.check(lib.lbc_bn_stats(None, pixels, C, bf, None, ctypes.byref(rows), None))
    part = torch.zeros((rows.value, 2, C), device=dev)
    _lib.check(lib.lbc_bn_stats(P(xd), pixels, C, bf, P(part), ctypes.byref(rows), _stream(xd)))
    assert torch.allclose(part[:, 0].sum(0).cpu(), x.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    gd, bd, rmd, rvd = gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    scale, shift, mean, invstd = (torch.empty(C, device=dev) for _ in range(4))
    _lib.check(lib.lbc_bn_finalize_stats(P(part), rows.value, C, pixels, P(gd), P(bd), P(rmd), P(rvd), P(nbt), 0.1, 1e-5, 1,
                                         P(scale), P(shift), P(mean), P(invstd), _stream(xd)))
    assert nbt.item() == 1
    assert torch.allclose(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-5) and torch.allclose(rvd.cpu(), rv_ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(mean.cpu(), x.mean((0, 2, 3)), rtol=1e-5, atol=1e-5)
    assert torch.allclose(invstd.cpu(), 1.0 / torch.sqrt(x.var((0, 2, 3), unbiased=False) + 1e-5), rtol=1e-5)
    buf, y = guarded((N, H, W, C), dev, dtype=_at(bf))
    _lib.check(lib.lbc_bn_apply_relu_add_fwd(P(xd), P(y), pixels, C, P(scale), P(shift), P(rd), None, None, 1, bf, _stream(xd)))
    check_guard(buf, y.numel())
    tol = 1e-5 if not bf else 2.0 ** -8
    assert relerr(nchw(y).float().cpu(), ref) < tol
    # residual that is itself BatchNorm'ed (downsample path), no ReLU
    rs, rt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ref2 = F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5) + (r * rs.view(1, -1, 1, 1) + rt.view(1, -1, 1, 1))
    rsd, rtd = rs.to(dev), rt.to(dev)
    _lib.check(lib.lbc_bn_apply_relu_add_fwd(P(xd), P(y), pixels, C, P(scale), P(shift), P(rd), P(rsd), P(rtd), 0, bf, _stream(xd)))
    assert relerr(nchw(y).float().cpu(), ref2) < tol


@pytest.mark.parametrize("shape", [BN_SHAPES[0]] + [BN_REAL[1]])
def test_bn_eval_forward(env, shape):
    dev, _ = env
    lib = _lib.get()
    N, C, H, W = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(shape, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    ref = F.batch_norm(x, rm.clone(), rv.clone(), gamma, beta, False, 0.1, 1e-5)
    xd = nhwc(x).to(dev)
    gd, bd, rmd, rvd = gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev)
    scale, shift = torch.empty(C, device=dev), torch.empty(C, device=dev)
    _lib.check(lib.lbc_bn_finalize_stats(None, 0, C, 0, P(gd), P(bd), P(rmd), P(rvd), None, 0.1, 1e-5, 0, P(scale), P(shift), None, None, _stream(xd)))
    assert torch.equal(rmd.cpu(), rm) and torch.equal(rvd.cpu(), rv)        # eval mode leaves the buffers alone
    y = torch.empty_like(xd)
    _lib.check(lib.lbc_bn_apply_relu_add_fwd(P(xd), P(y), N * H * W, C, P(scale), P(shift), None, None, None, 0, 0, _stream(xd)))
    assert relerr(nchw(y).cpu(), ref) < 1e-5


# ---------------------------------------------------------------------------------------------------------------
# BatchNorm2d backward fused with the ReLU backward of the layer above
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bf", [0, 1])
@pytest.mark.parametrize("variant", ["plain", "mask_tensor", "mask_affine", "first_channels"])
@pytest.mark.parametrize("shape", [(3, 64, 5, 7), (2, 128, 3, 5)] + [pytest.param((32, 128, 20, 48), marks=gpu), pytest.param((16, 640, 5, 12), marks=gpu)])
def test_bn_backward(env, shape, variant, bf):
    """dx, dgamma, dbeta of y = bn(x) in training mode, for the upstream gradient dz * (mask > 0):
    plain: no mask; mask_tensor: mask = relu output of the block (bn2 + identity); mask_affine: mask = bn(x) itself
    recomputed from x (conv1 -> bn1 -> relu, z1 never stored); first_channels: dx only for the first Cout channels."""
    dev, _ = env
    lib = _lib.get()
    N, C, H, W = shape
    g = torch.Generator().manual_seed(200 + C)
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3)
    dz = torch.randn(shape, generator=g)
    other = torch.randn(shape, generator=g)
    if bf:
        x, dz, other = rbf(x), rbf(dz), rbf(other)
    x.requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    y = F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5)
    Cout = C
    mask_t = mask_s = None
    if variant == "plain":
        out = y
    elif variant == "mask_tensor":
        out = F.relu(y + other)
        mask_t = out.detach()
    elif variant == "mask_affine":
        out = F.relu(y)
        mask_s = True
    else:
        out = y
        Cout = C // 2 if (C // 2) % 8 == 0 else C
    (out * dz).sum().backward()
    with torch.no_grad():
        mean = x.mean((0, 2, 3))
        invstd = 1.0 / torch.sqrt(x.var((0, 2, 3), unbiased=False) + 1e-5)
        scale = gamma * invstd
        shift = beta - mean * scale
    at = _at(bf)
    xd, dzd = nhwc(x.detach()).to(dev).to(at), nhwc(dz).to(dev).to(at)
    md = nhwc(mask_t).to(dev).to(at) if mask_t is not None else (xd if mask_s else None)
    sc, sh = (scale.to(dev), shift.to(dev)) if mask_s else (None, None)
    gout = torch.empty_like(dzd) if variant in ("mask_affine", "mask_tensor") else None
    gd, meand, invd = gamma.detach().to(dev), mean.to(dev), invstd.to(dev)
    dgamma, dbeta = torch.empty(C, device=dev), torch.empty(C, device=dev)
    buf, dx = guarded((N, H, W, Cout), dev, dtype=at)
    ws = torch.empty(lib.lbc_bn_bwd_workspace(C) // 4, device=dev)
    _lib.check(lib.lbc_bn_bwd(P(xd), P(dzd), P(md), P(sc), P(sh), P(gout), P(gd), P(meand), P(invd), N * H * W, C, Cout,
                              P(dgamma), P(dbeta), P(dx), P(ws), bf, _stream(xd)))
    check_guard(buf, dx.numel())
    # bf16 storage: the masked gradient g and dx are rounded once on store; the mask_affine decision relu(bn(x)) > 0 is taken on
    # the f32 affine of the bf16 x on both sides
    tol = 2e-5 if not bf else 2.0 ** -7
    assert relerr(nchw(dx).float().cpu(), x.grad[:, :Cout]) < tol
    assert relerr(dgamma.cpu(), gamma.grad) < (1e-4 if not bf else 1e-2) and relerr(dbeta.cpu(), beta.grad) < (1e-4 if not bf else 1e-2)


# ---------------------------------------------------------------------------------------------------------------
# bn1 -> relu -> maxpool(3,2,1) of the stem and its backward
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("bf", [0, 1])
@pytest.mark.parametrize("shape", [(2, 64, 8, 12), (1, 64, 6, 4), (3, 64, 2, 2)] + [pytest.param((8, 64, 80, 192), marks=gpu), pytest.param((4, 64, 96, 96), marks=gpu)])
def test_bn_relu_maxpool_fwd_bwd(env, shape, bf):
    dev, _ = env
    lib = _lib.get()
    N, C, H, W = shape
    g = torch.Generator().manual_seed(300 + H)
    y = torch.randn(shape, generator=g)
    if bf:
        y = rbf(y)
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.5
    mean, invstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    z = (y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).requires_grad_(True)
    p_ref = F.max_pool2d(F.relu(z), 3, 2, 1)
    dp = torch.randn(p_ref.shape, generator=g)
    if bf:
        dp = rbf(dp)
    (p_ref * dp).sum().backward()
    at = _at(bf)
    yd = nhwc(y).to(dev).to(at)
    sd, shd, md, ivd = scale.to(dev), shift.to(dev), mean.to(dev), invstd.to(dev)
    bufp, p = guarded((N, H // 2, W // 2, C), dev, dtype=at)
    idx = torch.full((N, H // 2, W // 2, C), 255, dtype=torch.uint8, device=dev)
    _lib.check(lib.lbc_maxpool3x3s2_fwd(P(yd), P(sd), P(shd), P(p), P(idx), N, H, W, C, bf, _stream(yd)))
    check_guard(bufp, p.numel())
    assert idx.max().item() <= 8
    pr = p_ref.detach()
    if bf:
        assert relerr(nchw(p).float().cpu(), pr) < 2.0 ** -8       # one rounding on store
    else:
        assert torch.allclose(nchw(p).cpu(), pr, rtol=1e-6, atol=1e-6)   # fma vs mul+add in the affine
    rows = ctypes.c_int(0)
    _lib.check(lib.lbc_maxpool3x3s2_bwd(None, None, None, None, None, None, None, None, None, ctypes.byref(rows), N, H, W, C, bf, None))
    part = torch.zeros((rows.value, 2, C), device=dev)
    dpd = nhwc(dp).to(dev).to(at)
    bufg, gg = guarded((N, H, W, C), dev, dtype=at)
    _lib.check(lib.lbc_maxpool3x3s2_bwd(P(dpd), P(idx), P(yd), P(sd), P(shd), P(md), P(ivd), P(gg), P(part), ctypes.byref(rows),
                                        N, H, W, C, bf, _stream(yd)))
    check_guard(bufg, gg.numel())
    gref = z.grad                    # gradient wrt the BatchNorm output, ReLU mask applied, arg-max routing
    got = nchw(gg).float().cpu()
    if bf:
        assert relerr(got, gref) < 2.0 ** -8
    else:
        assert torch.allclose(got, gref, rtol=0, atol=1e-6)    # sums of at most 4 window contributions in a different order
    xhat = (y - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    assert torch.allclose(part[:, 0].sum(0).cpu(), gref.sum((0, 2, 3)), rtol=1e-3, atol=1e-3 * gref.abs().sum((0, 2, 3)).max().item())
    assert torch.allclose(part[:, 1].sum(0).cpu(), (gref * xhat).sum((0, 2, 3)), rtol=1e-3, atol=1e-3 * (gref * xhat).abs().sum((0, 2, 3)).max().item())


# ---------------------------------------------------------------------------------------------------------------
# waypoint head
# ---------------------------------------------------------------------------------------------------------------
def _head_desc(dev, h_nhwc, N, OH, OW, bf, mean, invstd, gamma, beta, w, bias, cmd, keep):
    px, py = O.softmax_positions(OH, OW)
    d = _lib.HeadDesc()
    d.h = h_nhwc.data_ptr(); d.N = N; d.OH = OH; d.OW = OW; d.act_bf16 = bf
    pxd, pyd, cmdd = px.to(dev), py.to(dev), cmd.to(dev).contiguous()
    keep += [pxd, pyd, cmdd]
    for b in range(4):
        for name, t in (("mean", mean[b]), ("invstd", invstd[b]), ("gamma", gamma[b]), ("beta", beta[b]), ("w", w[b]), ("bias", bias[b])):
            td = t.detach().to(dev).contiguous()
            keep.append(td)
            getattr(d, name)[b] = td.data_ptr()
        d.pos_x[b] = pxd.data_ptr(); d.pos_y[b] = pyd.data_ptr()
    d.cmd = cmdd.data_ptr()
    return d, px, py


def _head_reference(h, gamma, beta, w, bias, px, py, cmd, train_stats):
    """4 x (BatchNorm2d(64) [training mode] -> Conv2d(64,5,1) -> SpatialSoftmax) -> stack -> select_branch (image.py:54-60,82-84)"""
    outs = []
    for b in range(4):
        z = F.batch_norm(h, None, None, gamma[b], beta[b], True, 0.1, 1e-5) if train_stats else h * gamma[b].view(1, -1, 1, 1) + beta[b].view(1, -1, 1, 1)
        outs.append(O.spatial_softmax(F.conv2d(z, w[b].view(5, 64, 1, 1), bias[b]), px, py))
    allb = torch.stack(outs, 1)
    return O.select_branch(allb, cmd), allb


@pytest.mark.parametrize("bf", [0, 1])
@pytest.mark.parametrize("shape", [(2, 6, 8), (3, 5, 11), (9, 4, 6)] + [pytest.param((4, 40, 96), marks=gpu), pytest.param((32, 40, 96), marks=gpu), pytest.param((4, 48, 48), marks=gpu)])
def test_head_forward_backward(env, shape, bf):
    """training-mode head: forward vs torch, backward (dh through the batch-statistics BatchNorm, dgamma, dbeta, dw, dbias)
    vs autograd; N = 9 exercises the pre-reduction of the per-sample partial rows"""
    dev, _ = env
    lib = _lib.get()
    N, OH, OW = shape
    g = torch.Generator().manual_seed(400 + N)
    h = torch.randn((N, 64, OH, OW), generator=g)
    if bf:
        h = rbf(h)
    h.requires_grad_(True)
    gamma = [(torch.rand(64, generator=g) + 0.5).requires_grad_(True) for _ in range(4)]
    beta = [(torch.randn(64, generator=g) * 0.3).requires_grad_(True) for _ in range(4)]
    w = [(torch.randn((5, 64), generator=g) * 0.4).requires_grad_(True) for _ in range(4)]
    bias = [torch.randn(5, generator=g).requires_grad_(True) for _ in range(4)]
    cmd = torch.eye(4)[torch.randint(0, 4, (N,), generator=g)]
    with torch.no_grad():
        mean = h.mean((0, 2, 3))
        invstd = 1.0 / torch.sqrt(h.var((0, 2, 3), unbiased=False) + 1e-5)
    keep = []
    hd = nhwc(h.detach()).to(dev).to(_at(bf))
    d, px, py = _head_desc(dev, hd, N, OH, OW, bf, [mean] * 4, [invstd] * 4, gamma, beta, w, bias, cmd, keep)
    # training mode: the four branches share ONE statistics vector (the executor passes the same pointer four times)
    for b in range(1, 4):
        d.mean[b] = d.mean[0]; d.invstd[b] = d.invstd[0]
    ref_sel, ref_all = _head_reference(h, gamma, beta, w, bias, px, py, cmd, True)
    ws = torch.zeros(lib.lbc_head_workspace(N) // 4, device=dev)
    pred_all, pred_sel = torch.empty((N, 4, 5, 2), device=dev), torch.empty((N, 5, 2), device=dev)
    _lib.check(lib.lbc_head_fwd(ctypes.byref(d), P(pred_all), P(pred_sel), P(ws), _stream(hd)))
    # bf16 activations: the MFMA head multiplies with the BatchNorm-folded 64 x 20 projection as a bf16 high + low pair (~16 bits): on
    # the same bf16 input it is as accurate as the f32 kernels.  (One bf16 copy of the weights -- round 3's form, removed in round 5 --
    # put a fixed 2^-9 relative perturbation on every logit term: 2e-2 in the waypoints at these sizes.)
    tol = 2e-5 if not bf else 1e-4
    err = max((pred_all.cpu() - ref_all).abs().max().item(), (pred_sel.cpu() - ref_sel).abs().max().item())
    assert err < tol, err
    d_all, d_sel = torch.randn((N, 4, 5, 2), generator=g), torch.randn((N, 5, 2), generator=g)
    ((ref_all * d_all).sum() + (ref_sel * d_sel).sum()).backward()
    dad, dsd = d_all.to(dev), d_sel.to(dev)
    bufh, dh = guarded((N, OH, OW, 64), dev, dtype=_at(bf))
    grads = {k: [torch.zeros(n, device=dev) for _ in range(4)] for k, n in (("dgamma", 64), ("dbeta", 64), ("dw", 320), ("dbias", 5))}
    arr = {k: (ctypes.c_void_p * 4)(*[t.data_ptr() for t in v]) for k, v in grads.items()}
    _lib.check(lib.lbc_head_bwd(ctypes.byref(d), P(pred_all), P(dad), P(dsd), P(dh), arr["dgamma"], arr["dbeta"], arr["dw"], arr["dbias"],
                                P(ws), _stream(hd)))
    check_guard(bufh, dh.numel())
    gt = 2e-4 if not bf else 5e-2
    assert relerr(nchw(dh).float().cpu(), h.grad) < gt
    for b in range(4):
        assert relerr(grads["dgamma"][b].cpu(), gamma[b].grad) < gt
        assert relerr(grads["dw"][b].cpu().view(5, 64), w[b].grad) < gt
        # the 1x1 conv's bias and the BatchNorm's beta both add a per-(branch, step) constant to the logit map, which
        # cancels inside the softmax: analytically zero gradients
        assert grads["dbias"][b].abs().max().item() < 1e-4 and bias[b].grad.abs().max().item() < 1e-4
        assert grads["dbeta"][b].abs().max().item() < (1e-4 if not bf else 1e-2) and beta[b].grad.abs().max().item() < 1e-4


@pytest.mark.parametrize("bf", [0, 1])
@pytest.mark.parametrize("shape", [(6, 8), pytest.param((40, 96), marks=gpu), pytest.param((48, 48), marks=gpu)])
def test_head_spatial_softmax_corner_known_answers(env, shape, bf):
    """reference common.py:192-201 (commented self-check): a logit map that is huge at one corner gives that corner's
    normalised coordinates: (-1,-1) top-left, (1,-1) top-right, (-1,1) bottom-left, (1,1) bottom-right; a uniform map gives (0,0)."""
    dev, _ = env
    lib = _lib.get()
    OH, OW = shape
    corners = [(0, 0, -1.0, -1.0), (0, OW - 1, 1.0, -1.0), (OH - 1, 0, -1.0, 1.0), (OH - 1, OW - 1, 1.0, 1.0)]
    N = 5
    h = torch.zeros((N, 64, OH, OW))
    for n, (yy, xx, _, _) in enumerate(corners):
        h[n, 3, yy, xx] = 1.0
    # identity BatchNorm (mean 0, invstd 1, gamma 1, beta 0); every step of every branch reads channel 3 with a large weight
    one, zero = torch.ones(64), torch.zeros(64)
    w = torch.zeros((5, 64))
    w[:, 3] = 64.0
    cmd = torch.eye(4)[torch.tensor([0, 1, 2, 3, 0])]
    keep = []
    hd = nhwc(h).to(dev).to(_at(bf))
    d, px, py = _head_desc(dev, hd, N, OH, OW, bf, [zero] * 4, [one] * 4, [one] * 4, [zero] * 4, [w] * 4, [torch.zeros(5)] * 4, cmd, keep)
    ws = torch.zeros(lib.lbc_head_workspace(N) // 4, device=dev)
    pred_all, pred_sel = torch.empty((N, 4, 5, 2), device=dev), torch.empty((N, 5, 2), device=dev)
    _lib.check(lib.lbc_head_fwd(ctypes.byref(d), P(pred_all), P(pred_sel), P(ws), _stream(hd)))
    pa, ps = pred_all.cpu(), pred_sel.cpu()
    for n, (_, _, ex, ey) in enumerate(corners):
        assert (pa[n, :, :, 0] - ex).abs().max().item() < 1e-6 and (pa[n, :, :, 1] - ey).abs().max().item() < 1e-6, (n, pa[n, 0, 0])
        assert (ps[n, :, 0] - ex).abs().max().item() < 1e-6 and (ps[n, :, 1] - ey).abs().max().item() < 1e-6
    assert pa[4].abs().max().item() < 1e-6      # uniform logits: the centre of the map


# ---------------------------------------------------------------------------------------------------------------
# stem: fused input pass + 7x7/2 convolution + weight gradient
# ---------------------------------------------------------------------------------------------------------------
MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (1, 7, 12, 12), (3, 3, 8, 40)] + [pytest.param((4, 3, 160, 384), marks=gpu), pytest.param((4, 7, 192, 192), marks=gpu)])
def test_stem_forward_and_weight_gradient(env, shape, mode):
    """mode 0: exact f32.  mode 1: bf16 padded image + bf16 MFMA operands, f32 output.  mode 2: bf16 output / gradient too."""
    dev, _ = env
    lib = _lib.get()
    N, C, H, W = shape
    g = torch.Generator().manual_seed(500 + C + H)
    u8 = torch.randint(0, 256, (N, H, W, C), generator=g, dtype=torch.uint8)
    x = (u8.float() / 255.0).permute(0, 3, 1, 2).contiguous()
    w = (torch.randn((64, C, 7, 7), generator=g) * (2.0 / (49 * 64)) ** 0.5).requires_grad_(True)
    xn = (x - MEAN) / STD if C == 3 else x
    xr = rbf(xn) if mode else xn
    wr = rbf(w.detach()).requires_grad_(True) if mode else w     # (a .to(bfloat16) inside the graph would round the gradient too)
    y_ref = F.conv2d(xr, wr, None, 2, 3)
    ref_tight = y_ref.detach()
    xbf = mode != 0
    pad_dt = torch.bfloat16 if xbf else torch.float32
    xp_a = torch.full((N, H + 6, W + 6, C), 7.0, dtype=pad_dt, device=dev)
    xp_b = torch.full((N, H + 6, W + 6, C), 7.0, dtype=pad_dt, device=dev)
    xd, u8d = x.to(dev), u8.to(dev)
    _lib.check(lib.lbc_nchw_to_input(P(xd), P(xp_a), int(xbf), N, C, H, W, int(C == 3), _stream(xd)))
    _lib.check(lib.lbc_u8nhwc_to_input(P(u8d), P(xp_b), int(xbf), N, C, H, W, int(C == 3), _stream(xd)))
    assert torch.equal(xp_a, xp_b)                     # the uint8 frames give bit-identical padded images
    inner = xp_a[:, 3:3 + H, 3:3 + W].float().cpu().permute(0, 3, 1, 2)
    assert (inner - xn).abs().max().item() < (2e-6 if not xbf else 2.0 ** -7)
    border = xp_a.float().clone()
    border[:, 3:3 + H, 3:3 + W] = 0
    assert border.abs().max().item() == 0.0            # 3-pixel zero border
    wd = w.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    at = torch.bfloat16 if mode == 2 else torch.float32
    rows = ctypes.c_int(0)
    _lib.check(lib.lbc_stem_fwd(None, None, None, None, ctypes.byref(rows), N, H, W, C, mode, None))
    st = torch.zeros((rows.value, 2, 64), device=dev)
    buf, y = guarded((N, H // 2, W // 2, 64), dev, dtype=at)
    _lib.check(lib.lbc_stem_fwd(P(xp_a), P(wd), P(y), P(st), ctypes.byref(rows), N, H, W, C, mode, _stream(xd)))
    check_guard(buf, y.numel())
    got = nchw(y).float().cpu()
    if mode == 0:
        assert relerr(got, ref_tight) < 1e-5
    else:
        # the padded image is the normalised frame rounded to bf16; the reference rounds the same f32 value (up to the last
        # f32 bit of the normalisation, which can straddle a bf16 boundary for a few pixels)
        assert relerr(got, ref_tight) < 2e-3 + (2.0 ** -8 if mode == 2 else 0)
        assert relerr(got, F.conv2d(xn, w.detach(), None, 2, 3)) < 2e-2
    assert torch.allclose(st[:, 0].sum(0).cpu(), got.sum((0, 2, 3)), rtol=2e-3, atol=2e-2)
    dy = torch.randn(y_ref.shape, generator=g)
    if mode == 2:
        dy = rbf(dy)
    y_ref.backward(rbf(dy) if mode else dy)
    dyd = nhwc(dy).to(dev).to(at)
    ws = torch.empty(lib.lbc_stem_wgrad_workspace(N, H, W, C) // 4 + 1, device=dev)
    bufw, dw = guarded((64, 7, 7, C), dev)
    _lib.check(lib.lbc_stem_wgrad(P(xp_a), P(dyd), P(dw), P(ws), N, H, W, C, mode, _stream(xd)))
    check_guard(bufw, dw.numel())
    assert relerr(dw.permute(0, 3, 1, 2).cpu(), wr.grad) < (2e-5 if mode == 0 else 5e-4)
