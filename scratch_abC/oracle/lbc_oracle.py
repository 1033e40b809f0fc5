"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the LbC sensorimotor hot path.

Plain torch fp32 on the CPU, functional style over a state_dict.  Nothing in the
product package imports this module; only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py do, and only as the checker / reported baseline.

Where the arithmetic lives: the reference (dotchen/LearningByCheating) delegates all
per-step arithmetic to third-party PyTorch (pinned torch==1.0.0, environment.yml:151-152)
-- nn.Conv2d / BatchNorm2d / ConvTranspose2d / MaxPool2d / ReLU / F.softmax / autograd /
optim.Adam.  This file restates the reference's *composition* of those ops (cited per
function) with the container's torch 2.10 CPU kernels supplying the arithmetic.

Pinning: oracle/make_golden.py runs this restatement against the real reference
classes imported from /root/reference (oracle/ref_shim.py) on seeded inputs and commits
the reference's outputs under tests/golden/; tests/test_oracle.py re-checks the
restatement against those fixtures wherever the suite runs.  The reference itself ships
no tests or golden vectors for this path.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = {"resnet18": [2, 2, 2, 2], "resnet34": [3, 4, 6, 3]}   # resnet.py:162-168 (BasicBlock variants)
STEPS = 5
COMMANDS = 4
PIXELS_PER_METER = 5
CROP_SIZE = 192


# ----------------------------------------------------------------------------------------
# state_dict layout (reference key order = module registration order)
# ----------------------------------------------------------------------------------------
def _bn_keys(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())]


def state_dict_layout(kind, backbone, height=None, width=None):
    """[(key, shape)] in the reference's state_dict order.
    image.py:22-62 / birdview.py:47-60 / resnet.py:95-146 / common.py:112-134."""
    cin = 3 if kind == "image" else 7
    if height is None:
        height, width = (160, 384) if kind == "image" else (192, 192)
    keys = [("conv.conv1.weight", (64, cin, 7, 7))] + _bn_keys("conv.bn1", 64)
    inpl = 64
    for li, nb in enumerate(LAYERS[backbone]):
        planes = 64 << li
        for bi in range(nb):
            p = "conv.layer%d.%d" % (li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            keys += [(p + ".conv1.weight", (planes, inpl, 3, 3))] + _bn_keys(p + ".bn1", planes)
            keys += [(p + ".conv2.weight", (planes, planes, 3, 3))] + _bn_keys(p + ".bn2", planes)
            if stride != 1 or inpl != planes:
                keys += [(p + ".downsample.0.weight", (planes, inpl, 1, 1))] + _bn_keys(p + ".downsample.1", planes)
            inpl = planes
    keys += [("conv.fc.weight", (1000, 512)), ("conv.fc.bias", (1000,))]      # resnet.py:111-112, never used in forward
    chans = [640, 256, 128, 64]
    for i in range(3):
        keys += _bn_keys("deconv.%d" % (3 * i), chans[i])
        keys += [("deconv.%d.weight" % (3 * i + 1), (chans[i], chans[i + 1], 3, 3)),
                 ("deconv.%d.bias" % (3 * i + 1), (chans[i + 1],))]
    hw = (height // 4) * (width // 4)
    for b in range(COMMANDS):
        p = "location_pred.%d" % b
        keys += _bn_keys(p + ".0", 64)
        keys += [(p + ".1.weight", (STEPS, 64, 1, 1)), (p + ".1.bias", (STEPS,)),
                 (p + ".2.pos_x", (hw,)), (p + ".2.pos_y", (hw,))]
    return keys


def softmax_positions(map_h, map_w):
    """pos_x / pos_y buffers of SpatialSoftmax (common.py:127-134).  The reference builds them with
    np.meshgrid(linspace(-1,1,height), linspace(-1,1,width)) where it passes height=map_w and
    width=map_h (image.py:52,58), which yields a (map_h, map_w) grid flattened row-major."""
    pos_x, pos_y = np.meshgrid(np.linspace(-1.0, 1.0, map_w), np.linspace(-1.0, 1.0, map_h))
    return (torch.from_numpy(pos_x.reshape(map_h * map_w)).float(), torch.from_numpy(pos_y.reshape(map_h * map_w)).float())


def make_state_dict(kind, backbone, seed, height=None, width=None, trained_like=True):
    """Seeded, non-trivial weights (BN affine/running stats perturbed so that every term of the
    arithmetic is exercised).  torch CPU generators are platform independent for a given torch build,
    so fixtures only need the seed."""
    if height is None:
        height, width = (160, 384) if kind == "image" else (192, 192)
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    layout = state_dict_layout(kind, backbone, height, width)
    bn_prefixes = {k[:-len(".running_mean")] for k, _ in layout if k.endswith(".running_mean")}
    for key, shape in layout:
        prefix, leaf = key.rsplit(".", 1)
        if leaf == "num_batches_tracked":
            sd[key] = torch.tensor(0, dtype=torch.long)
        elif leaf in ("pos_x", "pos_y"):
            px, py = softmax_positions(height // 4, width // 4)
            sd[key] = px if leaf == "pos_x" else py
        elif leaf == "running_var":
            sd[key] = torch.rand(shape, generator=g) + 0.5 if trained_like else torch.ones(shape)
        elif leaf == "running_mean":
            sd[key] = torch.randn(shape, generator=g) * 0.1 if trained_like else torch.zeros(shape)
        elif prefix in bn_prefixes and leaf == "weight":
            sd[key] = torch.rand(shape, generator=g) * 0.5 + 0.75 if trained_like else torch.ones(shape)
        elif len(shape) == 1:   # BN bias / conv bias / fc bias
            sd[key] = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 2:
            sd[key] = torch.randn(shape, generator=g) * 0.01
        else:
            if key.startswith("deconv"):
                fan = shape[0] * 9 / 4.0     # each output pixel of a stride-2 transposed conv sees ~9/4 taps
            else:
                fan = shape[1] * shape[2] * shape[3]
            sd[key] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan)
    return sd


def checksum(sd):
    """order-dependent fingerprint of a state_dict (float64 sums), to prove two sites built the same weights"""
    acc = 0.0
    for i, (k, v) in enumerate(sd.items()):
        acc += (i + 1) * float(v.double().sum()) + float(v.double().abs().sum())
    return acc


# ----------------------------------------------------------------------------------------
# forward (functional)
# ----------------------------------------------------------------------------------------
# Emulation of the executor's optional "bf16 MFMA operand" precision (lbc_net_desc.precision = 1): every operand of a
# trunk/decoder convolution GEMM -- activations, weights and, in the backward pass, output gradients -- is rounded to
# bf16 (RNE) right before the multiply; accumulation, tensors and everything else stay f32.  The stem convolution is
# rounded the same way; the head's 1x1 convolution is not.  Off by default (the reference arithmetic is f32).
MFMA_BF16 = False


def _rbf(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _RoundGradBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rbf(g)


# Emulation of precision = 2 on top of MFMA_BF16: every activation tensor the executor keeps in HBM (raw convolution
# outputs, block outputs, the pooled stem output, the velocity-concatenated map, decoder outputs) is rounded to bf16 where
# it is stored, and so is the gradient that flows back into it (activation gradients are stored as bf16 too).  The
# executor's BatchNorm statistics come from the f32 accumulators; here they see the rounded tensor (a 2^-9-relative,
# zero-mean difference).
ACT_BF16 = False


class _RoundBothBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rbf(x)

    @staticmethod
    def backward(ctx, g):
        return _rbf(g)


def _st(x):
    """a tensor the executor stores in HBM"""
    return _RoundBothBF16.apply(x) if ACT_BF16 else x


def _conv(x, w, bias, stride, pad):
    if MFMA_BF16:
        return _RoundGradBF16.apply(F.conv2d(_rbf(x), _rbf(w), None, stride, pad)) + (0 if bias is None else bias.view(1, -1, 1, 1))
    return F.conv2d(x, w, bias, stride, pad)


def _deconv(x, w, bias):
    if MFMA_BF16:
        return _RoundGradBF16.apply(F.conv_transpose2d(_rbf(x), _rbf(w), None, 2, 1, 1)) + bias.view(1, -1, 1, 1)
    return F.conv_transpose2d(x, w, bias, 2, 1, 1)


BN_MOMENTUM = 0.1   # torch default, used everywhere in the reference; tests may set it to 1.0 to calibrate running stats


def _bn(sd, prefix, x, train):
    """nn.BatchNorm2d: momentum 0.1, eps 1e-5; training mode updates running stats in sd."""
    if train:
        sd[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], train, BN_MOMENTUM, 1e-5)


def calibrate_running_stats(sd, kind, backbone, x, velocity, command):
    """Overwrite the running statistics of `sd` with the batch statistics of (x, velocity): what a trained network's
    buffers look like.  (Seeded random running stats let eval-mode activations grow to ~5e4, an ill-conditioned case.)"""
    global BN_MOMENTUM
    old, BN_MOMENTUM = BN_MOMENTUM, 1.0
    try:
        with torch.no_grad():
            policy_forward(sd, kind, backbone, x, velocity, command, True)
    finally:
        BN_MOMENTUM = old
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k].zero_()
    return sd


def _relu(x, frozen, key):
    """nn.ReLU -- or, with `frozen` masks, the linear map the implementation under test applied at this site"""
    return F.relu(x) if frozen is None else x * frozen[key].to(x.dtype)


def _relu_maxpool(x, frozen):
    """relu -> MaxPool2d(3, 2, 1) (resnet.py:150-152).  Frozen: out[n,c,oy,ox] = x[n,c,iy,ix] * positive[n,c,oy,ox] with the window
    element (iy, ix) = (2 oy - 1 + tap // 3, 2 ox - 1 + tap % 3) chosen by frozen["conv.maxpool.idx"] (relu and max commute)."""
    if frozen is None:
        return F.max_pool2d(F.relu(x), 3, 2, 1)
    tap = frozen["conv.maxpool.idx"].long()
    n, c, oh, ow = tap.shape
    oy = torch.arange(oh).view(1, 1, oh, 1)
    ox = torch.arange(ow).view(1, 1, 1, ow)
    iy, ix = 2 * oy - 1 + tap // 3, 2 * ox - 1 + tap % 3
    assert int(iy.min()) >= 0 and int(ix.min()) >= 0 and int(iy.max()) < x.shape[2] and int(ix.max()) < x.shape[3]
    picked = torch.gather(x.reshape(n, c, -1), 2, (iy * x.shape[3] + ix).reshape(n, c, -1)).reshape(n, c, oh, ow)
    return picked * frozen["conv.maxpool"].to(x.dtype)


def trunk(sd, backbone, x, train, taps=None, frozen=None):
    """ResNet.forward (resnet.py:148-159) with BasicBlock.forward (resnet.py:38-54)."""
    x = _st(_conv(x, sd["conv.conv1.weight"], None, 2, 3))
    x = _st(_relu_maxpool(_bn(sd, "conv.bn1", x, train), frozen))
    if taps is not None:
        taps["pool"] = x
    inpl = 64
    for li, nb in enumerate(LAYERS[backbone]):
        planes = 64 << li
        for bi in range(nb):
            p = "conv.layer%d.%d" % (li + 1, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            identity = x
            out = _st(_conv(x, sd[p + ".conv1.weight"], None, stride, 1))
            out = _relu(_bn(sd, p + ".bn1", out, train), frozen, p + ".bn1")
            out = _st(_conv(out, sd[p + ".conv2.weight"], None, 1, 1))
            out = _bn(sd, p + ".bn2", out, train)
            if stride != 1 or inpl != planes:
                identity = _bn(sd, p + ".downsample.1", _st(_conv(x, sd[p + ".downsample.0.weight"], None, stride, 0)), train)
            x = _st(_relu(out + identity, frozen, p))
            inpl = planes
        if taps is not None:
            taps["layer%d" % (li + 1)] = x
    return x


def spatial_softmax(feature, pos_x, pos_y):
    """SpatialSoftmax.forward (common.py:136-152), data_format NCHW, temperature 1."""
    n, c, h, w = feature.shape
    if feature.dtype == torch.bfloat16:      # only under the autocast comparator of the tests: CUDA autocast runs softmax in float32
        feature = feature.float()
    weight = F.softmax(feature.reshape(-1, h * w), dim=-1)
    ex = torch.sum(pos_x * weight, dim=1, keepdim=True)
    ey = torch.sum(pos_y * weight, dim=1, keepdim=True)
    return torch.cat([ex, ey], 1).view(-1, c, 2)


def select_branch(branches, one_hot):
    """common.py:29-35: sum_b one_hot[n,b] * branches[n,b,...]"""
    return torch.sum(one_hot[:, :, None, None] * branches, dim=1)


def policy_forward(sd, kind, backbone, x, velocity, command, train, taps=None, frozen=None):
    """ImagePolicyModelSS.forward (image.py:64-89) / BirdViewPolicyModelSS.forward (birdview.py:62-79).
    Returns (location_pred (N,5,2), location_preds (N,4,5,2)).

    frozen (test device, not in the reference): {site: 0/1 mask (N,C,H,W)} for every ReLU ("conv.layerL.B.bn1", "conv.layerL.B",
    "deconv.2/5/8") plus "conv.maxpool" (positive mask of the pooled map) and "conv.maxpool.idx" (chosen window tap) -- the
    branch decisions of an implementation under test.  The network is piecewise linear in its activations; with the decisions
    frozen, this function and that implementation differentiate the SAME smooth function, so their gradients may be compared
    at round-off level instead of "up to the occasional kink flip" (tests/test_model.py::_frozen_gradient_check)."""
    if kind == "image":
        mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        x = (x - mean) / std                                                     # common.py:108-109
    h = trunk(sd, backbone, x, train, taps, frozen)
    b, c, kh, kw = h.shape
    vel = velocity[..., None, None, None].repeat((1, 128, kh, kw))              # image.py:77
    h = _st(torch.cat((h, vel), dim=1))
    for i in range(3):                                                          # image.py:37-47
        h = _bn(sd, "deconv.%d" % (3 * i), h, train)
        h = _deconv(h, sd["deconv.%d.weight" % (3 * i + 1)], sd["deconv.%d.bias" % (3 * i + 1)])
        h = _st(_relu(h, frozen, "deconv.%d" % (3 * i + 2)))
    if taps is not None:
        taps["decoder"] = h
    preds = []
    for br in range(COMMANDS):                                                  # image.py:54-60,82
        p = "location_pred.%d" % br
        z = _bn(sd, p + ".0", h, train)
        z = F.conv2d(z, sd[p + ".1.weight"], sd[p + ".1.bias"])
        preds.append(spatial_softmax(z, sd[p + ".2.pos_x"], sd[p + ".2.pos_y"]))
    preds = torch.stack(preds, dim=1)
    return select_branch(preds, command), preds


def one_hot(x, num_digits=4, start=1):
    """bird_view/utils/train_utils.py:33-40"""
    idx = torch.clamp(x.long()[:, None] - start, 0, num_digits - 1)
    y = torch.zeros(x.shape[0], num_digits)
    y.scatter_(1, idx, 1)
    return y


# ----------------------------------------------------------------------------------------
# phase-1 / phase-0 / bird-view losses
# ----------------------------------------------------------------------------------------
def phase1_unproject(camera_locations, w=384, h=160, fov=90, world_y=1.4, fixed_offset=4.0):
    """CoordConverter.__call__ of training/train_image_phase1.py:43-64 (differentiable)."""
    img = torch.tensor([float(w), float(h)])
    loc = (camera_locations + 1) * img / 2
    f = w / (2 * np.tan(fov * np.pi / 360))
    xt = (loc[..., 0] - w / 2) / f
    yt = (loc[..., 1] - h / 2) / f
    world_z = world_y / yt
    world_x = world_z * xt
    mx = world_x * PIXELS_PER_METER + CROP_SIZE / 2
    my = CROP_SIZE - world_z * PIXELS_PER_METER + fixed_offset * PIXELS_PER_METER
    return torch.stack([mx, my], dim=-1)


def phase1_loss(pred_map_locations, teacher_locations):
    """LocationLoss.forward of train_image_phase1.py:66-70 -> per-sample loss (N,)"""
    p = pred_map_locations / (0.5 * CROP_SIZE) - 1
    return torch.mean(torch.abs(p - teacher_locations), dim=(1, 2, 3))


def phase0_project(map_locations, w=384, h=160, fov=90, world_y=1.4, fixed_offset=4.0):
    """CoordConverter.__call__ of train_image_phase0.py:67-79 with _project_image_xy (:54-65).
    cv2.projectPoints with zero rvec/tvec and no distortion is the pinhole map u = f X/Z + cx, v = f Y/Z + cy
    evaluated in float64."""
    t = map_locations.detach().cpu().numpy()
    t = (t + 1) * CROP_SIZE / 2
    t[:, :, 1] = CROP_SIZE - t[:, :, 1]
    t[:, :, 0] -= CROP_SIZE / 2
    t = t / PIXELS_PER_METER
    t[:, :, 1] += fixed_offset
    f = w / (2 * np.tan(fov * np.pi / 360))
    X = t[..., 0].astype(np.float64)
    Z = t[..., 1].astype(np.float64)
    u = np.clip(f * X / Z + w / 2, 0, w)
    v = np.clip(f * world_y / Z + h / 2, 0, h)
    return torch.FloatTensor(np.stack([u, v], -1))


def phase0_loss(pred_locations, image_locations, w=384, h=160):
    """LocationLoss.forward of train_image_phase0.py:86-89 -> per-sample loss (N,)"""
    img = torch.tensor([float(w), float(h)])
    loc = image_locations / (0.5 * img) - 1
    return torch.mean(torch.abs(pred_locations - loc), dim=(1, 2))


def phase2_weight(pred_location_cam, teacher_location):
    """resampling weight of train_image_phase2.py:203-206: get_weight (phase2_utils.py:50-59) on the selected-branch
    prediction, unprojected to the map frame and normalised"""
    learner = phase1_unproject(pred_location_cam) / (0.5 * CROP_SIZE) - 1.0
    decay = torch.tensor([0.7 ** i for i in range(5)])
    xy_bias = torch.tensor([0.7, 0.3])
    return torch.mean((torch.abs(learner - teacher_location) * xy_bias).sum(dim=-1) * decay, dim=-1)


def repeat(a, repeats, dim=0):
    """np.repeat-style interleave used for --batch_aug (train_image_phase1.py:131-154): [1,2,3] -> [1,1,2,2,3,3]"""
    return torch.repeat_interleave(a, repeats, dim=dim)


def birdview_loss(pred_location, gt_location, size=192):
    """LocationLoss(choice='l1') of training/train_birdview.py:33-54 -> per-sample loss (N,)"""
    gt = gt_location / (0.5 * size) - 1.0
    return torch.mean(torch.abs(pred_location - gt), dim=(1, 2))


# ----------------------------------------------------------------------------------------
# the phase-1 training step (train_image_phase1.py:157-229), used for gradient parity and as
# the reported CPU baseline
# ----------------------------------------------------------------------------------------
def as_params(sd):
    """clone a state_dict into leaf tensors requiring grad (floating point entries that the reference
    registers as nn.Parameter)"""
    out = OrderedDict()
    for k, v in sd.items():
        is_param = v.dtype.is_floating_point and not (k.endswith("running_mean") or k.endswith("running_var")
                                                      or k.endswith("pos_x") or k.endswith("pos_y"))
        out[k] = v.clone().requires_grad_(True) if is_param else v.clone()
    return out


def phase1_step_loss(student, teacher, backbone_s, backbone_t, rgb, birdview, speed, command):
    """teacher forward (eval, no grad) -> student forward (train) -> unprojection -> L1 over all 4 branches"""
    with torch.no_grad():
        _, teac_all = policy_forward(teacher, "birdview", backbone_t, birdview, speed, command, False)
    pred, pred_all = policy_forward(student, "image", backbone_s, rgb, speed, command, True)
    loss = phase1_loss(phase1_unproject(pred_all), teac_all)
    return loss, pred, pred_all, teac_all
