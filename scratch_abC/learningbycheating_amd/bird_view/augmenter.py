"""Colour augmentation of the RGB frames on the GPU (reference bird_view/augmenter.py, which builds imgaug==0.2.8
pipelines that run per sample in the CPU dataloader workers, bird_view/utils/datasets/image_lmdb.py:137-140,220).

Same strategy names and the same strength schedules (every factor is the reference's formula of the running image
counter); the operators themselves run as HIP kernels over a whole uint8 batch (csrc/data.hip, C ABI
lbc_augment_rgb_u8).  Per image the host draws, exactly as iaa.Sequential(random_order=True) of iaa.Sometimes(p, op) does:
which operators fire (probability `frequency`), their order, their magnitudes from the reference's ranges, and whether each
is per-channel (probability `color`).  imgaug's own random stream is NOT reproduced (the package is absent here and its
draws depend on numpy's Mersenne Twister call order): the restatement is distribution-level, and the per-pixel noise /
dropout masks come from a counter-based hash on the device (tests/test_data.py holds the numpy twin of every operator).

    aug = super_hard(image_iteration)        # reference call shape: getattr(augmenter, name)(counter)
    aug.augment_batch(rgb_u8)                # (N,160,384,3) uint8 device tensor, in place
"""
import ctypes

import numpy as np
import torch

from .. import _lib

BLUR, NOISE, COARSE_DROPOUT, DROPOUT, ADD, MULTIPLY, CONTRAST = range(7)


class Recipe:
    """strengths of one strategy at a given image counter + the sampling of per-image parameters"""

    def __init__(self, name, frequency, color, ops):
        self.name, self.frequency, self.color, self.ops = name, float(min(max(frequency, 0.0), 1.0)), float(min(max(color, 0.0), 1.0)), ops

    def sample(self, n, rng, height=160, width=384):
        """-> (ctypes array of n lbc_aug_params, any_blur)"""
        arr = (_lib.AugParams * n)()
        any_blur = False
        for i in range(n):
            p = arr[i]
            fired = [op for op in self.ops if rng.random_sample() < self.frequency]       # iaa.Sometimes(frequency, .)
            rng.shuffle(fired)                                                            # random_order=True
            p.n_ops = len(fired)
            p.blur_pos = len(fired)
            p.seed = int(rng.randint(0, 2 ** 31 - 1))
            for k, op in enumerate(fired):
                p.order[k] = op
                pc = int(rng.random_sample() < self.color)                                 # per_channel=color_factor
                r = self.ops[op]
                if op == BLUR:
                    p.blur_sigma = float(rng.uniform(*r))
                    p.blur_pos = k
                    any_blur = True
                elif op == NOISE:
                    p.noise_scale, p.noise_pc = float(rng.uniform(*r)), pc
                elif op == COARSE_DROPOUT:
                    p.coarse_p, p.coarse_pc = float(rng.uniform(*r["p"])), pc
                    sp = rng.uniform(*r["size_percent"])                                   # iap.FromLowerResolution: min_size 4
                    p.coarse_h, p.coarse_w = max(4, int(round(height * sp))), max(4, int(round(width * sp)))
                elif op == DROPOUT:
                    p.dropout_p, p.dropout_pc = float(rng.uniform(*r)), pc
                elif op in (ADD, MULTIPLY, CONTRAST):
                    if op == ADD:       # iaa.Add on uint8 draws integers
                        vals = [float(rng.randint(int(np.floor(r[0])), int(np.floor(r[1])) + 1)) for _ in range(3 if pc else 1)]
                    else:
                        vals = [float(rng.uniform(*r)) for _ in range(3 if pc else 1)]
                    dst = {ADD: p.add, MULTIPLY: p.multiply, CONTRAST: p.contrast}[op]
                    for c in range(3):
                        dst[c] = vals[c if pc else 0]
        return arr, any_blur


class BatchAugmenter:
    def __init__(self, recipe, seed=0):
        self.recipe = recipe
        self.rng = np.random.RandomState(seed)
        self._scratch = None

    def augment_batch(self, rgb_u8, params=None):
        """rgb_u8: contiguous uint8 (N,H,W,3) device tensor, augmented in place; returns it"""
        if rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 4 or rgb_u8.shape[3] != 3 or not rgb_u8.is_contiguous():
            raise RuntimeError("augment_batch: expected a contiguous uint8 (N,H,W,3) tensor")
        _lib.require_device(rgb_u8)
        n, h, w, _ = rgb_u8.shape
        arr, any_blur = params if params is not None else self.recipe.sample(n, self.rng, h, w)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        dev = host.to(rgb_u8.device)
        scratch = None
        if any_blur:
            if self._scratch is None or self._scratch.numel() < n * h * w * 3 or self._scratch.device != rgb_u8.device:
                self._scratch = torch.empty(n * h * w * 3, dtype=torch.float32, device=rgb_u8.device)
            scratch = self._scratch
        _lib.check(_lib.get().lbc_augment_rgb_u8(_lib.ptr(rgb_u8), _lib.ptr(dev), _lib.ptr(scratch), n, h, w, int(any_blur),
                                                 _lib.stream_for(rgb_u8)), "augment_rgb_u8")
        if rgb_u8.is_cuda:
            dev.record_stream(torch.cuda.current_stream(rgb_u8.device))
        return rgb_u8


def _dropout_factor(iteration):
    return 0.198667 + (0.03856658 - 0.198667) / (1 + (iteration / 196416.6) ** 1.863486)


def _standard_ops(it, blur_div, add_div, mul_pos_div, mul_neg_div, con_div):
    d = _dropout_factor(it)
    return {
        BLUR: (0.0, 0.5 + 0.5 * it / blur_div),
        NOISE: (0.0, d),                                      # (the reference passes the dropout factor as the noise scale)
        COARSE_DROPOUT: {"p": (0.0, d), "size_percent": (0.08, 0.2)},
        DROPOUT: (0.0, d),
        ADD: (-(10 + 10 * it / add_div), 10 + 10 * it / add_div),
        MULTIPLY: (1 - 0.91 * it / mul_neg_div, 1 + 2.5 * it / mul_pos_div),
        CONTRAST: (1 - 0.5 * it / con_div, 1 + 0.5 * it / con_div),
    }


def super_hard(image_iteration):
    """reference bird_view/augmenter.py:227-279 (the default --augment of training/train_image_phase{0,1}.py)"""
    it = image_iteration / 32.0
    return Recipe("super_hard", min(0.05 + it / 50000.0, 1.0), it / 100000.0, _standard_ops(it, 100000.0, 100000.0, 200000.0, 500000.0, 500000.0))


def medium(image_iteration):
    """reference bird_view/augmenter.py:16-67 (without its Grayscale operator, which the default recipe does not use either)"""
    it = image_iteration / (32 * 1.5)
    return Recipe("medium", 0.05 + it / 1000000.0, it / 1000000.0, _standard_ops(it, 100000.0, 150000.0, 500000.0, 500000.0, 500000.0))


def medium_harder(image_iteration):
    """reference bird_view/augmenter.py:174-225: medium's strengths on a counter that advances 1.5x faster (iteration = images / 32)"""
    it = image_iteration / 32.0
    return Recipe("medium_harder", 0.05 + it / 1000000.0, it / 1000000.0, _standard_ops(it, 100000.0, 150000.0, 500000.0, 500000.0, 500000.0))


def custom(image_iteration):
    """reference bird_view/augmenter.py:282-330: no coarse dropout / contrast, Add(-30, 30) shared, Multiply(0.9, 1.3) per channel"""
    it = image_iteration / 32.0
    d = _dropout_factor(it)
    ops = {BLUR: (0.0, 0.5 + 0.5 * it / 20000.0), NOISE: (0.0, d), DROPOUT: (0.0, d), ADD: (-30.0, 30.0), MULTIPLY: (0.9, 1.3)}
    r = Recipe("custom", min(0.05 + it / 50000.0, 1.0), it / 100000.0, ops)
    return r


STRATEGIES = {"super_hard": super_hard, "medium": medium, "medium_harder": medium_harder, "custom": custom}


def get(name):
    if name in (None, "None"):
        return None
    return STRATEGIES[name]
