"""The LbC training step on the native executor, without autograd bookkeeping:
    teacher forward (eval) -> student forward (train) -> loss kernel -> staged backward
    (+ bucketed RCCL all-reduce) -> fused Adam
= reference training/train_image_phase1.py:174-205 (phase 1), train_image_phase0.py:163-189
(phase 0) and train_birdview.py:116-128 (bird-view behaviour cloning)."""
import ctypes

import torch

from .. import _lib
from ..engine import PolicyEngine
from ..optim import FusedAdam
from ..parallel import StageAllReducer

CAMERA = dict(w=384.0, h=160.0, fov=90.0, world_y=1.4, fixed_offset=4.0, pixels_per_meter=5.0, crop_size=192.0)


def camera_struct(**kw):
    c = dict(CAMERA)
    c.update(kw)
    return _lib.Camera(c["w"], c["h"], c["fov"], c["world_y"], c["fixed_offset"], c["pixels_per_meter"], c["crop_size"])


class NativeTrainer:
    """phase: 1 (student vs teacher, all branches, map space), 0 (student vs teacher, selected branch,
    image space), 'birdview' (privileged agent vs ground-truth waypoints), 'l1_all' (all branches vs
    given normalised targets; used to warm-start synthetic benchmarks below the horizon)."""

    def __init__(self, student, teacher, batch, image_shape, device, phase=1, lr=1e-4, world_size=1, group=None, camera=None, grad_dtype=None,
                 sync_bn=False, teacher_shape=(7, 192, 192)):
        self.student, self.teacher, self.phase, self.batch, self.world = student, teacher, phase, batch, world_size
        self.device = device
        student.train()
        self.eng = student.engine((batch,) + tuple(image_shape), device, max_batch=batch, with_grads=True)
        self.teng = None
        if teacher is not None:
            teacher.eval()
            self.teng = teacher.engine((batch,) + tuple(teacher_shape), device, max_batch=batch, with_grads=False)
            # the privileged teacher is frozen (train_image_phase1.py:244-248: loaded, eval(), never stepped): its bf16 weight copies and
            # folded BatchNorm affines are derived on the first forward only
            self.teng.set_frozen(True)
            self._teacher_versions = self._versions(teacher)
        self.cam = camera or camera_struct()
        self.opt = FusedAdam(list(student.named_parameters()), self.eng.grad_views, lr=lr)
        self.reducer = StageAllReducer(self.eng.grad_flat, self.eng.grad_spans, group, grad_dtype=grad_dtype)   # grad_dtype: see parallel.py
        self.sync_bn = bool(sync_bn and world_size > 1)
        if sync_bn and world_size > 1:
            # BatchNorm over the global batch (not in the reference: it trains 256 images on one device, which is what this
            # restores for 8 x 32).  On a GPU the reductions run on the library's own RCCL communicator (`group` only carries
            # its id); the torch.distributed transport of the CPU emulator gets a process group of its own
            import torch.distributed as dist
            rccl = torch.device(device).type == "cuda" and dist.get_backend(group) == "nccl"
            self.eng.set_sync_bn(group if (rccl or group is not None) else dist.new_group())
        self.loss = torch.zeros(batch, dtype=torch.float32, device=device)
        self.dpred_all = torch.zeros((batch, 4, 5, 2), dtype=torch.float32, device=device)
        self.dpred_sel = torch.zeros((batch, 5, 2), dtype=torch.float32, device=device)
        self.nstages = PolicyEngine.num_stages()
        # the frozen teacher's forward is independent of the student's: it runs on a side stream, which fills the GPU at
        # small per-GPU batches (both networks launch kernels far smaller than the chip there)
        self.side = torch.cuda.Stream(device=device) if (teacher is not None and torch.device(device).type == "cuda") else None
        self.overlap_teacher = True      # False: one stream (per-kernel timing of an instrumented step stays meaningful)

    @staticmethod
    def _versions(module):
        """torch's in-place version counters of a module's tensors: an optimizer step, an EMA update or a `p.copy_()` on the frozen teacher
        moves them (writes through `p.data` do not: after those call `trainer.teng.invalidate()`)"""
        return tuple(t._version for t in list(module.parameters()) + list(module.buffers()))

    def _loss(self, kind, pred, target, rows, dpred):
        n = pred.shape[0]
        _lib.check(_lib.get().lbc_loss(kind, ctypes.byref(self.cam), _lib.ptr(pred), _lib.ptr(target), n, rows,
                                       1.0 / (n * self.world), _lib.ptr(self.loss), _lib.ptr(dpred), _lib.stream_for(pred)), "loss")

    def step(self, x, speed, command, birdview=None, target=None, update=True, train_mode=True, on_forward=None):
        """x: student input, float32 (N,C,H,W) in [0,1] or the dataset's uint8 (N,H,W,C) frames; command one-hot (N,4);
        returns the per-sample loss (device tensor).  update=False: forward + loss only.  train_mode=False: the student runs
        in eval mode (running statistics, no buffer update) -- the reference's validation pass (train_image_phase1.py:162-165,256).
        on_forward (parity tests): called with the trainer after the student's forward, before the loss and the backward."""
        if not train_mode and update:
            raise ValueError("an eval-mode step cannot update (backward through running-statistics BatchNorm is not implemented)")
        n = x.shape[0]
        # the executor takes raw pointers to dense tensors; a permuted / sliced view is packed first (the reference's
        # nn.Module accepts any strides)
        x, speed, command = x.contiguous(), speed.contiguous(), command.contiguous()
        if birdview is not None:
            birdview = birdview.contiguous()
        if self.phase in (0, 1):
            if not getattr(self.teng, "_frozen", False):        # (somebody ran the teacher through its module API since: the promise is ours again)
                self.teng.set_frozen(True)
            v = self._versions(self.teacher)
            if v != self._teacher_versions:                     # the "frozen" teacher was written in place: derive its weight copies again
                self.teng.invalidate()
                self._teacher_versions = v
            if self.side is not None and self.overlap_teacher:
                main = torch.cuda.current_stream(self.device)
                self.side.wait_stream(main)                      # inputs (and last step's use of the teacher outputs) are ordered before
                with torch.cuda.stream(self.side):
                    t_sel, t_all = self.teng.forward(birdview, speed, command, False)
                t_sel.record_stream(main); t_all.record_stream(main)
            else:
                t_sel, t_all = self.teng.forward(birdview, speed, command, False)
            self.last_teacher = (t_sel, t_all)
        p_sel, p_all = self.eng.forward(x, speed, command, bool(train_mode))
        if self.phase in (0, 1) and self.side is not None and self.overlap_teacher:
            torch.cuda.current_stream(self.device).wait_stream(self.side)      # the loss reads the teacher's waypoints
        self.last_pred = (p_sel, p_all)
        if on_forward is not None:
            on_forward(self)
        d_sel = d_all = None
        if self.phase == 1:
            self._loss(1, p_all, t_all, 20, self.dpred_all); d_all = self.dpred_all[:n]
        elif self.phase == 0:
            self._loss(0, p_sel, t_sel, 5, self.dpred_sel); d_sel = self.dpred_sel[:n]
        elif self.phase == "birdview":
            self._loss(2, p_sel, target, 5, self.dpred_sel); d_sel = self.dpred_sel[:n]
        elif self.phase == "l1_all":
            self._loss(3, p_all, target, 20, self.dpred_all); d_all = self.dpred_all[:n]
        else:
            raise ValueError(self.phase)
        if update:
            for st in range(self.nstages):
                self.eng.backward(d_sel, d_all, st)
                self.reducer.launch(st)
                if self.sync_bn:
                    # two communicators (the buckets' and the BatchNorm rows') must meet in ONE order on every rank: kernels of two
                    # RCCL communicators that become resident in different orders on two devices can wait for each other forever.
                    # With synchronized BatchNorm the next stage's rows therefore queue behind this stage's bucket (no overlap of the
                    # bucket with the backward in this mode; local BatchNorm -- the default -- has a single communicator and keeps it)
                    self.reducer.fence()
            self.reducer.wait()
            self.opt.step()
        return self.loss[:n]
