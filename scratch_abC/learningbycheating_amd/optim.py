"""Multi-tensor Adam on the HIP kernel (csrc/adam.hip) -- torch.optim.Adam semantics
(reference call sites training/train_image_phase{0,1}.py:231,252, lr 1e-4)."""
import ctypes

import numpy as np
import torch

from . import _lib

CHUNK = 32768


class FusedAdam:
    """Adam over (param, grad) pairs whose physical memory is dense (any stride permutation);
    grads are the engine's flat-buffer views, so p, g, m, v share one element order."""

    def __init__(self, named_params, grads, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.names = [n for n, _ in named_params if n in grads]
        params = dict(named_params)
        # moments in the gradients' element order, every tensor starting on a 64-element (256-byte) boundary: adam_k moves
        # 16 bytes per lane, so p, g, m and v of every chunk must be 16-byte aligned (the engine's flat gradient buffer pads
        # the same way; the 5-element head biases would otherwise misalign everything behind them)
        from .engine import _pad
        total = sum(_pad(params[n].numel()) for n in self.names)
        dev = params[self.names[0]].device
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        rows, off = [], 0
        self.offsets = {}
        for n in self.names:
            p, g = params[n].data, grads[n]
            assert p.stride() == g.stride() and p.dtype == torch.float32
            assert p.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0, "adam: %s is not 16-byte aligned" % n
            self.offsets[n] = (off, p.numel())
            for c in range(0, p.numel(), CHUNK):
                k = min(CHUNK, p.numel() - c)
                rows.append((p.data_ptr() + 4 * c, g.data_ptr() + 4 * c, self.exp_avg.data_ptr() + 4 * (off + c),
                             self.exp_avg_sq.data_ptr() + 4 * (off + c), k))
            off += _pad(p.numel())
        table = np.zeros(len(rows), dtype=[("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i4"), ("pad", "<i4")])
        for i, r in enumerate(rows):
            table[i] = r + (0,)
        assert table.itemsize == ctypes.sizeof(_lib.AdamChunk)
        self.nchunks = len(rows)
        _lib.get().lbc_adam_profile_elems(sum(params[n].numel() for n in self.names))     # (books the launch profiler's 28 bytes per element)
        self.table = torch.from_numpy(table.view(np.uint8).copy()).to(dev)
        self._keep = (params, grads)

    def step(self):
        self.step_count += 1
        _lib.check(_lib.get().lbc_adam_step(_lib.ptr(self.table), self.nchunks, self.lr, self.betas[0], self.betas[1], self.eps,
                                            self.weight_decay, self.step_count, _lib.stream_for(self.table)), "adam_step")

    def state_of(self, name):
        off, n = self.offsets[name]
        return self.exp_avg[off:off + n], self.exp_avg_sq[off:off + n]
