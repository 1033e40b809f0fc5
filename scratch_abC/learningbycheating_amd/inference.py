"""Batch-1 inference session: the consumer on the other side of the checkpoint (reference bird_view/models/image.py:124-139
ImageAgent.run_step / birdview.py:107-118 BirdViewAgent.run_step, called once per simulator tick by benchmark_agent.py:36-38).

The reference converts the camera frame with ToTensor on the host (uint8 HWC -> float CHW / 255), copies 737 KB to the
device and runs ~150 eager kernels.  Here the frame crosses PCIe as the 184 KB uint8 it is, /255 + normalisation + NHWC
repack are the first kernel, every BatchNorm is folded into its convolution (eval mode), and the whole forward -- about 70
launches at batch 1 -- is captured ONCE into a hipGraph and replayed per tick, so the per-step host cost is one
hipGraphLaunch instead of one launch per kernel.
"""
import numpy as np
import torch

from . import _lib


class PolicySession:
    """model: ImagePolicyModelSS / BirdViewPolicyModelSS with its weights loaded.  run_step(frame, speed, command) -> (5, 2)
    numpy waypoints in normalised [-1, 1] camera / map coordinates (what the agents feed their controllers)."""

    def __init__(self, model, device, use_graph=True):
        self.model = model.to(device).eval()
        self.device = torch.device(device)
        c = model.input_channel
        h, w = (160, 384) if c == 3 else (192, 192)
        self.eng = model.engine((1, c, h, w), self.device, max_batch=1, with_grads=False)
        self.frame = torch.zeros((1, h, w, c), dtype=torch.uint8, device=self.device)
        # Host staging is ordinary pageable memory on purpose.  Measured on the MI355X box (scripts/diag_latency.py): a pinned
        # (hipHostMalloc) buffer that the GPU has read since the CPU last wrote it costs 3.2 ms to rewrite (184 KB) and 5.7 ms
        # for the next H2D -- the pages behave like migrating managed memory -- so a per-tick pinned staging buffer turned
        # a 0.9 ms forward into a 9 ms tick; a plain copy_ from pageable memory (the runtime's own staging) costs ~30 us.
        self.h_small = torch.zeros(5, dtype=torch.float32)                  # speed + one-hot command
        self.small = torch.zeros(5, dtype=torch.float32, device=self.device)
        self.speed = self.small[:1]
        self.command = self.small[1:].view(1, 4)
        self.graph = None
        for _ in range(2):                                                  # warm-up outside the capture (lazy allocations, zero page)
            self.out_sel, self.out_all = self.eng.forward(self.frame, self.speed, self.command, False)
        if use_graph and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out_sel, self.out_all = self.eng.forward(self.frame, self.speed, self.command, False)

    def run_step(self, frame, speed, command):
        """frame: uint8 (H,W,C) numpy array or tensor as the simulator delivers it; command: 1..4 (reference one_hot[command - 1])"""
        f = torch.as_tensor(np.ascontiguousarray(frame) if isinstance(frame, np.ndarray) else frame)
        if f.dtype != torch.uint8 or tuple(f.shape) != tuple(self.frame.shape[1:]):
            raise ValueError("run_step: expected a uint8 frame of shape %s, got %s %s" % (tuple(self.frame.shape[1:]), f.dtype, tuple(f.shape)))
        self.h_small.zero_()
        self.h_small[0] = float(speed)
        self.h_small[1 + min(max(int(command) - 1, 0), 3)] = 1.0
        self.frame[0].copy_(f)                  # 184 KB / 258 KB, pageable -> device
        self.small.copy_(self.h_small)          # speed and one-hot command in one 20-byte copy
        if self.graph is not None:
            self.graph.replay()
        else:
            self.out_sel, self.out_all = self.eng.forward(self.frame, self.speed, self.command, False)
        return self.out_sel[0].cpu().numpy()    # 40 bytes back; .cpu() waits for the stream
