"""Shared pieces of the policy models (reference bird_view/models/common.py), MI355X-native.

ResnetBase keeps the reference constructor; `PolicyBase` adds the executor plumbing used by
ImagePolicyModelSS / BirdViewPolicyModelSS: one torch.autograd.Function spans the whole
network, so `loss.backward()` in a caller's training loop reaches the HIP backward kernels
exactly once.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from ...engine import PolicyEngine, is_channels_last_4d
from .resnet import get_resnet

CROP_SIZE = 192
MAP_SIZE = 320


def select_branch(branches, one_hot):
    """reference common.py:29-35 (tiny host-side tensor op; the model forward fuses it into the head kernel)"""
    shape = branches.size()
    for i, s in enumerate(shape[2:]):
        one_hot = torch.stack([one_hot for _ in range(s)], dim=i + 2)
    return torch.sum(one_hot * branches, dim=1)


class ResnetBase(nn.Module):
    """reference common.py:69-83"""

    def __init__(self, backbone, input_channel=3, bias_first=True, pretrained=False):
        super().__init__()
        conv, c = get_resnet(backbone, input_channel=input_channel, bias_first=bias_first, pretrained=pretrained)
        self.conv = conv
        self.c = c
        self.backbone = backbone
        self.input_channel = input_channel
        self.bias_first = bias_first


class SpatialSoftmax(nn.Module):
    """Owner of the pos_x / pos_y buffers (reference common.py:112-134).  NB the reference passes
    (height=map_w, width=map_h) (image.py:52,58); the resulting grid is (map_h, map_w) row-major.
    The soft-argmax itself is fused with the 1x1 conv in csrc/head.hip."""

    def __init__(self, height, width, channel, temperature=None, data_format="NCHW"):
        super().__init__()
        if temperature or data_format != "NCHW":
            raise NotImplementedError("only temperature=None / NCHW are live in the reference (common.py:123,141 are broken)")
        self.data_format, self.height, self.width, self.channel = data_format, height, width, channel
        self.temperature = 1.0
        pos_x, pos_y = np.meshgrid(np.linspace(-1.0, 1.0, self.height), np.linspace(-1.0, 1.0, self.width))
        self.register_buffer("pos_x", torch.from_numpy(pos_x.reshape(self.height * self.width)).float())
        self.register_buffer("pos_y", torch.from_numpy(pos_y.reshape(self.height * self.width)).float())

    def forward(self, feature):
        raise RuntimeError("SpatialSoftmax runs fused inside the HIP head kernel; call the policy model instead")


def spatial_softmax_decoder():
    """BN -> ConvTranspose2d(3,2,1,1) -> ReLU x 3 (reference image.py:37-47 / birdview.py:34-44)"""
    return nn.Sequential(
        nn.BatchNorm2d(640), nn.ConvTranspose2d(640, 256, 3, 2, 1, 1), nn.ReLU(True),
        nn.BatchNorm2d(256), nn.ConvTranspose2d(256, 128, 3, 2, 1, 1), nn.ReLU(True),
        nn.BatchNorm2d(128), nn.ConvTranspose2d(128, 64, 3, 2, 1, 1), nn.ReLU(True))


class _PolicyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, need_grad, image, velocity, command, *params):
        ctx.set_materialize_grads(False)
        eng = module._engine_for(image, need_grad)   # (grad mode is always off inside Function.forward: decided by the caller)
        pred_sel, pred_all = eng.forward(image, velocity, command, module.training)
        ctx.module, ctx.eng, ctx.train = module, eng, module.training
        ctx.generation = eng.generation            # the activations backward() needs live in the engine's ONE workspace
        return pred_sel, pred_all

    @staticmethod
    def backward(ctx, d_sel, d_all):
        if not ctx.train:
            raise RuntimeError("backward through an eval-mode (running-statistics) forward is not implemented")
        if d_sel is None and d_all is None:
            return (None,) * (5 + len(ctx.module._param_names))
        eng = ctx.eng
        if eng.generation != ctx.generation:
            raise RuntimeError(
                "backward through a stale forward: the executor keeps the activations of the LAST forward only (one workspace per "
                "module and input size), and %d more forward(s) ran on this module since the one being differentiated. Call "
                "backward() before the next forward of the same module (the reference's training loops do), or use a second "
                "module instance for interleaved forwards." % (eng.generation - ctx.generation))
        eng.backward(None if d_sel is None else d_sel.contiguous().float(),
                     None if d_all is None else d_all.contiguous().float())
        grads = []
        for n in ctx.module._param_names:
            v = eng.grad_views.get(n)
            grads.append(None if v is None else v.clone(memory_format=torch.preserve_format))
        return (None, None, None, None, None) + tuple(grads)


class PolicyBase(ResnetBase):
    """Executor plumbing shared by the two policy models."""
    _normalize = False
    #: "fp32" (default, the parity path: exact-f32 MFMA everywhere); "bf16" = mixed precision as in BASELINE.json config 3
    #: (convolution MFMA operands and the activations / activation gradients stored in HBM are bf16; f32 accumulation,
    #: f32 master weights, gradients, BatchNorm statistics, soft-argmax, loss and Adam); "bf16_mfma" = only the MFMA
    #: operands are rounded to bf16, every tensor stays f32.  Set before the first forward.
    precision = "fp32"

    def _finish_init(self):
        # 4-D weights live in channels_last memory order: logical shapes (= checkpoint shapes) stay
        # (O,I,kh,kw) / (I,O,kh,kw) while HBM holds [O][kh][kw][I] / [I][kh][kw][O] for the kernels.
        for p in self.parameters():
            if p.dim() == 4:
                p.data = p.data.contiguous(memory_format=torch.channels_last)
        self._engines = {}
        self._param_names = [n for n, _ in self.named_parameters()]

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        for p in self.parameters():
            if p.dim() == 4 and not is_channels_last_4d(p.data):
                p.data = p.data.contiguous(memory_format=torch.channels_last)
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        for eng in getattr(self, "_engines", {}).values():     # (values copied in place: a frozen engine must derive its weight copies again)
            eng.invalidate()
        return out

    def _arch(self):
        return {"resnet18": 18, "resnet34": 34}[self.backbone]

    def _engine_for(self, image, with_grads):
        n, c, h, w = image.shape
        prec = {"fp32": 0, "bf16_mfma": 1, "bf16": 2}[self.precision]
        key = (h, w, str(image.device), prec)
        eng = self._engines.get(key)
        if eng is None or eng.max_batch < n:
            eng = PolicyEngine(self._arch(), self.input_channel, h, w, self._normalize, max(n, 1), image.device, prec)
            self._engines[key] = eng
        # "frozen" is a promise of whoever holds the engine (NativeTrainer for its teacher), not a property of the cached engine: anyone who
        # comes through the module API gets an engine that derives its weight copies and folded BatchNorm affines from the tensors as they are
        if getattr(eng, "_frozen", False):
            eng.set_frozen(False)
        tensors = dict(self.named_parameters())
        tensors.update(dict(self.named_buffers()))
        eng.bind({k: v.data for k, v in tensors.items() if k in set(eng.names)}, with_grads or eng.grad_flat is not None,
                 param_order=self._param_names)
        return eng

    def engine(self, image_shape, device, max_batch=None, with_grads=True):
        """Executor for a given input shape (used by the native training loops / bench)."""
        n, c, h, w = image_shape
        probe = torch.empty((max_batch or n, c, h, w), device=device, dtype=torch.float32)
        return self._engine_for(probe, with_grads)

    def _run(self, x, velocity, command):
        _lib.require_device(x)
        if x.dim() != 4 or x.shape[1] != self.input_channel:
            raise ValueError("expected input of shape (N,%d,H,W), got %s" % (self.input_channel, tuple(x.shape)))
        n = x.shape[0]
        velocity = velocity.reshape(-1)
        if velocity.shape[0] != n or tuple(command.shape) != (n, 4):
            raise ValueError("velocity must be (N,) and command (N,4) one-hot; got %s / %s" % (tuple(velocity.shape), tuple(command.shape)))
        x = x.contiguous().float()
        velocity = velocity.to(x.device).contiguous().float()
        command = command.to(x.device).contiguous().float()
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _PolicyFunction.apply(self, need_grad, x, velocity, command, *params)
