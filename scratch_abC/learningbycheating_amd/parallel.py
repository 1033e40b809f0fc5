"""Data parallelism for the LbC training step: one process per GPU, gradients all-reduced
(sum; the loss is pre-scaled by 1/world_size) over RCCL/xGMI in per-stage buckets that are
launched while the remaining backward stages still execute.  The reference has no
distributed code at all (single `cuda` device, training/train_image_phase1.py:297).

Buckets are contiguous ranges of the executor's flat gradient buffer, one per backward stage
(head+decoder | layer4 | layer3 | layer2 | layer1 | stem), so no packing copies are needed.
"""
import torch
import torch.distributed as dist

STAGE_PREFIXES = [("deconv.", "location_pred."), ("conv.layer4.",), ("conv.layer3.",), ("conv.layer2.",), ("conv.layer1.",),
                  ("conv.conv1.", "conv.bn1.")]


def stage_ranges(grad_spans):
    """[(start, end)] element ranges of the flat gradient buffer, one per backward stage.  grad_spans: name -> (offset,
    padded element count) as laid out by PolicyEngine.bind (every tensor on a 256-byte boundary; pads are zero)."""
    out = []
    for prefixes in STAGE_PREFIXES:
        spans = [grad_spans[n] for n in grad_spans if n.startswith(prefixes)]
        lo = min(o for o, _ in spans)
        hi = max(o + c for o, c in spans)
        assert sum(c for _, c in spans) == hi - lo, "stage parameters must be contiguous in the flat gradient buffer"
        out.append((lo, hi))
    return out


class StageAllReducer:
    """grad_dtype: None / torch.float32 = the f32 buckets are reduced in place (bit-reproducible sum order per RCCL ring);
    torch.bfloat16 = every bucket is cast into a bf16 staging buffer on the communication stream, reduced there (half the
    bytes on the xGMI links: 46 MB instead of 92 MB per step for ResNet-34) and cast back into the f32 gradient buffer --
    the usual compressed-gradient data parallelism of a bf16 run with f32 master weights (BASELINE.json config 3)."""

    def __init__(self, grad_flat, grad_spans, group=None, force=False, grad_dtype=None):
        self.flat = grad_flat
        self.ranges = stage_ranges(grad_spans)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())   # force: exercise the comm path on one rank (tests)
        self.cuda = grad_flat.is_cuda
        self.comm = torch.cuda.Stream(device=grad_flat.device) if self.cuda and self.active else None
        self.pending = []
        self.staging = None
        if self.active and grad_dtype is not None and grad_dtype != grad_flat.dtype:
            self.staging = torch.empty(grad_flat.numel(), dtype=grad_dtype, device=grad_flat.device)

    def _reduce(self, lo, hi):
        bucket = self.flat[lo:hi]
        if self.staging is None:
            self.pending.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        # compressed bucket: cast, reduce, cast back -- all in stream order (the cast back must see the reduced values)
        st = self.staging[lo:hi]
        st.copy_(bucket)
        dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)
        bucket.copy_(st)

    def launch(self, stage):
        """call right after enqueueing backward stage `stage` on the current stream"""
        if not self.active:
            return
        lo, hi = self.ranges[stage]
        if self.comm is not None:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                self._reduce(lo, hi)
        else:
            self._reduce(lo, hi)

    def participants(self):
        """how many ranks the buckets' communicator really joins: a one from every rank, summed on the path the buckets take (same
        group, same stream, the staging dtype if there is one).  bench.py reports it as `comm_ranks` -- read back from the communicator,
        not from the launcher's environment"""
        if not self.active:
            return 1
        one = torch.ones(1, dtype=self.staging.dtype if self.staging is not None else self.flat.dtype, device=self.flat.device)
        if self.comm is not None:
            self.comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm):
                dist.all_reduce(one, op=dist.ReduceOp.SUM, group=self.group)
            torch.cuda.current_stream().wait_stream(self.comm)
        else:
            dist.all_reduce(one, op=dist.ReduceOp.SUM, group=self.group)
        return int(round(float(one.float().item())))

    def fence(self):
        """the current stream waits for everything launched on the communication stream so far (gloo / CPU: launches are synchronous
        or waited for here)"""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)


def broadcast_module(module, src=0, group=None):
    """identical initial weights/buffers on every rank"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        d = t.data
        if d.dim() == 4 and not d.is_contiguous() and d.is_contiguous(memory_format=torch.channels_last):
            d = d.permute(0, 2, 3, 1)       # the channels_last weights as the plain-contiguous view of the same memory
        dist.broadcast(d, src, group=group)
