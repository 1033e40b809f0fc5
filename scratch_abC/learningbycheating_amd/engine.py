"""Python handle on the native network executor (csrc/engine.cpp, C ABI lbc_net_*)."""
import ctypes

import torch

from . import _lib


GRAD_ALIGN = 64      # elements (256 bytes)
SYNC_FLOATS = 1536   # Net::kSyncFloats
COMM_ID_BYTES = 128  # LBC_COMM_ID_BYTES
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)   # lbc_allreduce_fn


def _pad(n):
    return (n + GRAD_ALIGN - 1) // GRAD_ALIGN * GRAD_ALIGN


def is_channels_last_4d(t):
    """True when a 4-D tensor's memory order is [d0][d2][d3][d1] (or the distinction is void)."""
    if t.dim() != 4:
        return True
    o, i, kh, kw = t.shape
    want = (kh * kw * i, 1, kw * i, i)
    st = t.stride()
    return all(t.shape[d] == 1 or st[d] == want[d] for d in range(4))


class PolicyEngine:
    """Binds an nn.Module's parameters/buffers (by state_dict name) to an lbc_net plan."""

    def __init__(self, arch, in_channels, height, width, normalize, max_batch, device, precision=0):
        lib = _lib.get()
        self.desc = _lib.NetDesc(arch, in_channels, height, width, int(normalize), max_batch, int(precision))
        self.precision = int(precision)
        h = ctypes.c_void_p()
        _lib.check(lib.lbc_net_create(ctypes.byref(self.desc), ctypes.byref(h)), "net_create")
        self.handle = h
        self.device = device
        self.max_batch = max_batch
        n = lib.lbc_net_num_tensors(h)
        self.names, self.kinds, self.shapes = [], [], []
        buf = ctypes.create_string_buffer(256)
        kind, ndim = ctypes.c_int(), ctypes.c_int()
        shape = (ctypes.c_int * 4)()
        for i in range(n):
            _lib.check(lib.lbc_net_tensor_info(h, i, buf, 256, ctypes.byref(kind), ctypes.byref(ndim), shape), "tensor_info")
            self.names.append(buf.value.decode())
            self.kinds.append(kind.value)
            self.shapes.append(tuple(shape[k] for k in range(ndim.value)))
        self.workspace = torch.empty(lib.lbc_net_workspace_bytes(h), dtype=torch.uint8, device=device)
        self._bound_key = None
        self.generation = 0          # forwards run on this engine's workspace so far (autograd checks it, see models/common.py)
        self.last_batch = 0
        self.grad_flat = None
        self.grad_views = {}
        self._keep = None
        self._sync = None

    def __del__(self):
        try:
            if self.handle:
                self._release_sync()
                _lib.get().lbc_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- binding ---------------------------------------------------------------------
    def bind(self, tensors, with_grads, param_order=None):
        """tensors: dict name -> tensor (parameters and buffers).  When with_grads, one flat fp32
        gradient buffer is (re)used and every parameter gets a view of it in the parameter's own
        memory order; param_order (names) fixes the layout of the flat buffer (bucket order)."""
        key = (tuple(tensors[n].data_ptr() for n in self.names), bool(with_grads))
        if key == self._bound_key:
            return
        for n, shape in zip(self.names, self.shapes):
            t = tensors[n]
            if tuple(t.shape) != shape:
                raise RuntimeError("engine.bind: %s has shape %s, expected %s" % (n, tuple(t.shape), shape))
            if t.dim() == 4 and not is_channels_last_4d(t):
                raise RuntimeError("engine.bind: %s must be in channels_last memory order" % n)
            if t.device != self.workspace.device:
                raise RuntimeError("engine.bind: %s lives on %s, engine on %s" % (n, t.device, self.workspace.device))
        nt = len(self.names)
        tp = (ctypes.c_void_p * nt)(*[tensors[n].data_ptr() for n in self.names])
        gp = None
        if with_grads:
            pnames = [n for n, k in zip(self.names, self.kinds) if k == 0]
            order = [n for n in (param_order or pnames) if n in set(pnames)]
            assert set(order) == set(pnames)
            # every tensor starts on a GRAD_ALIGN-element (256-byte) boundary of the flat buffer: the fused Adam and the
            # all-reduce buckets address it with 16-byte vector accesses (the 5-element head biases would otherwise
            # misalign everything behind them); the pad elements stay zero and travel with their stage's bucket
            total = sum(_pad(tensors[n].numel()) for n in order)
            if self.grad_flat is None or self.grad_flat.numel() != total:
                self.grad_flat = torch.zeros(total, dtype=torch.float32, device=self.workspace.device)
            self.grad_views, self.grad_offsets, self.grad_spans, off = {}, {}, {}, 0
            for n in order:
                t = tensors[n]
                self.grad_views[n] = torch.as_strided(self.grad_flat, t.shape, t.stride(), off)
                self.grad_offsets[n] = (off, t.numel())
                self.grad_spans[n] = (off, _pad(t.numel()))
                off += _pad(t.numel())
            gp = (ctypes.c_void_p * nt)(*[self.grad_views[n].data_ptr() if k == 0 else 0 for n, k in zip(self.names, self.kinds)])
        _lib.check(_lib.get().lbc_net_bind(self.handle, _lib.ptr(self.workspace), tp, gp), "net_bind")
        self._bound_key = key
        self._keep = tensors

    # ---- execution --------------------------------------------------------------------
    def forward(self, image, velocity, command, train):
        """image: float32 (N,C,H,W) in [0,1] (the reference signature) or uint8 (N,H,W,C) frames as the dataset stores them"""
        d = self.desc
        if image.dim() != 4:
            raise RuntimeError("engine.forward: image must be 4-D, got %s" % (tuple(image.shape),))
        n = image.shape[0]
        if not 1 <= n <= self.max_batch:
            raise RuntimeError("engine.forward: batch %d outside [1, %d]" % (n, self.max_batch))
        # the C ABI takes raw pointers: everything it will dereference is validated here (dtype, layout, extent, device)
        if image.dtype == torch.uint8:
            want = (n, d.H, d.W, d.in_channels)
            fn = _lib.get().lbc_net_forward_u8
        elif image.dtype == torch.float32:
            want = (n, d.in_channels, d.H, d.W)
            fn = _lib.get().lbc_net_forward
        else:
            raise RuntimeError("engine.forward: image must be float32 (N,C,H,W) or uint8 (N,H,W,C), got %s" % image.dtype)
        if tuple(image.shape) != want or not image.is_contiguous():
            raise RuntimeError("engine.forward: image must be a contiguous %s tensor of shape %s (the engine's plan), got %s with strides %s"
                               % (image.dtype, want, tuple(image.shape), image.stride()))
        for name, t, shape in (("velocity", velocity, (n,)), ("command", command, (n, 4))):
            if t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous():
                raise RuntimeError("engine.forward: %s must be a contiguous float32 tensor of shape %s, got %s %s"
                                   % (name, shape, t.dtype, tuple(t.shape)))
        for name, t in (("image", image), ("velocity", velocity), ("command", command)):
            if t.device != self.workspace.device:
                raise RuntimeError("engine.forward: %s lives on %s, the engine on %s" % (name, t.device, self.workspace.device))
        # (synchronized BatchNorm: every all-reduced row of sums carries this rank's batch size behind it, and the finalize kernels
        #  divide by the summed count -- ranks may run different batch sizes, and no rank enters a collective the others might skip)
        pred_sel = torch.empty((n, 5, 2), dtype=torch.float32, device=image.device)
        pred_all = torch.empty((n, 4, 5, 2), dtype=torch.float32, device=image.device)
        self._check(fn(self.handle, n, int(train), _lib.ptr(image), _lib.ptr(velocity), _lib.ptr(command),
                       _lib.ptr(pred_sel), _lib.ptr(pred_all), _lib.stream_for(image)), "net_forward")
        self.generation += 1
        self.last_batch = n
        return pred_sel, pred_all

    def backward(self, d_sel, d_all, stage=-1):
        ref = d_sel if d_sel is not None else d_all
        for name, t, shape in (("d_sel", d_sel, (self.last_batch, 5, 2)), ("d_all", d_all, (self.last_batch, 4, 5, 2))):
            if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.workspace.device):
                raise RuntimeError("engine.backward: %s must be a contiguous float32 %s tensor on %s (the last forward ran %d samples), got %s %s on %s"
                                   % (name, shape, self.workspace.device, self.last_batch, t.dtype, tuple(t.shape), t.device))
        self._check(_lib.get().lbc_net_backward(self.handle, _lib.ptr(d_sel), _lib.ptr(d_all), stage, _lib.stream_for(ref)), "net_backward")

    def _check(self, rc, what):
        """_lib.check, with the exception a SyncBN all-reduce callback caught (it must not unwind through the C frames) as the cause"""
        sync = getattr(self, "_sync", None)
        err = sync.get("error") if sync else None
        if rc != 0 and err is not None:
            sync["error"] = None
            raise RuntimeError("lbc_hip %s failed (%d): the synchronized-BatchNorm all-reduce raised %s: %s"
                               % (what, rc, type(err).__name__, err)) from err
        _lib.check(rc, what)

    def set_frozen(self, frozen=True):
        """the caller promises not to touch parameters / buffers (a frozen teacher): eval-mode forwards then derive the bf16 weight
        copies and the folded BatchNorm affines once instead of per forward (lbc_net_set_frozen); any re-bind derives them again"""
        self._frozen = bool(frozen)
        _lib.check(_lib.get().lbc_net_set_frozen(self.handle, int(self._frozen)), "net_set_frozen")

    def invalidate(self):
        """the bound tensors were rewritten in place (load_state_dict): whatever a frozen engine derived from them is derived again"""
        _lib.check(_lib.get().lbc_net_set_frozen(self.handle, int(getattr(self, "_frozen", False))), "net_set_frozen")

    # ---- introspection (parity tests) ------------------------------------------------------
    def activations(self):
        """{name: tensor view} of the activations the last training-mode forward left in the workspace (lbc_net_activation_info):
        (N, H, W, C) views in the stored element type (float32, bfloat16 in precision 2, uint8 for the max-pool arg-max taps)."""
        lib = _lib.get()
        out = {}
        buf = ctypes.create_string_buffer(256)
        off, hwc, eb = ctypes.c_size_t(), (ctypes.c_int * 3)(), ctypes.c_int()
        n = self.last_batch
        for i in range(lib.lbc_net_num_activations(self.handle)):
            _lib.check(lib.lbc_net_activation_info(self.handle, i, buf, 256, ctypes.byref(off), hwc, ctypes.byref(eb)), "activation_info")
            dt = {4: torch.float32, 2: torch.bfloat16, 1: torch.uint8}[eb.value]
            rows = 1 if (hwc[0] == 1 and hwc[1] == 1) else n          # per-channel vectors ("...bn1.scale") have no batch axis
            numel = rows * hwc[0] * hwc[1] * hwc[2]
            raw = self.workspace[off.value: off.value + numel * eb.value]
            out[buf.value.decode()] = raw.view(dt).view(rows, hwc[0], hwc[1], hwc[2])
        return out

    # ---- synchronized BatchNorm (data parallelism) ---------------------------------------
    def set_sync_bn(self, group=None, enable=True, native=None):
        """Every training-mode BatchNorm uses the statistics of the global batch (lbc_net_set_sync_bn): before each finalize
        the native executor has one row of per-channel sums all-reduced over the data-parallel group, in stream order.
        native (default on a GPU under the nccl backend): the library's own RCCL communicator (lbc_comm_*) -- its id is created on group rank 0 and
        broadcast over `group`; each reduction is one ncclAllReduce enqueued from C.  native=False (default on the CPU
        emulator / gloo): the executor calls back into torch.distributed on `group`; give that a process group of its own,
        on the gradient buckets' communicator the small reductions of the next backward stage would queue behind a bucket."""
        import torch.distributed as dist
        lib = _lib.get()
        self._release_sync()
        if not enable:
            _lib.check(lib.lbc_net_set_sync_bn(self.handle, None, None, 1, None, 0), "net_set_sync_bn")
            return
        world = dist.get_world_size(group)
        dev = self.workspace.device
        buf = torch.zeros(SYNC_FLOATS, dtype=torch.float32, device=dev)
        # the per-channel sums travel with this rank's batch size behind them (csrc/engine.cpp Net::sync_rows)
        self._sync_group = (group, world)
        if native is None:      # (gloo ranks may share one GPU in self-tests: RCCL refuses two ranks on one device)
            native = dev.type == "cuda" and dist.get_backend(group) == "nccl"
        if native:
            ident = torch.zeros(COMM_ID_BYTES, dtype=torch.uint8)
            rank = dist.get_rank(group)
            if rank == 0:
                _lib.check(lib.lbc_comm_unique_id(_lib.ptr(ident)), "comm_unique_id")
            wire = ident.to(dev) if dist.get_backend(group) == "nccl" else ident
            dist.broadcast(wire, dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = wire.cpu()
            comm = ctypes.c_void_p()
            with torch.cuda.device(dev):       # ncclCommInitRank binds the current device
                _lib.check(lib.lbc_comm_create(_lib.ptr(ident), rank, world, ctypes.byref(comm)), "comm_create")
            fn = ctypes.cast(lib.lbc_comm_allreduce_f32, ctypes.c_void_p)
            _lib.check(lib.lbc_net_set_sync_bn(self.handle, fn, comm, world, _lib.ptr(buf), SYNC_FLOATS), "net_set_sync_bn")
            self._sync = {"comm": comm, "buf": buf, "error": None}
            return
        state = {"buf": buf, "group": group, "error": None}

        def reduce_row(ctx, ptr, count, stream):
            try:
                assert ptr == buf.data_ptr() and 0 < count <= SYNC_FLOATS
                dist.all_reduce(buf[:count], op=dist.ReduceOp.SUM, group=group)    # on the current stream = the executor's
                return 0
            except Exception as e:       # must not propagate through the C frames
                state["error"] = e
                return 1

        state["fn"] = ALLREDUCE_FN(reduce_row)       # the C side holds this function pointer and the buffer
        _lib.check(lib.lbc_net_set_sync_bn(self.handle, ctypes.cast(state["fn"], ctypes.c_void_p), None, world, _lib.ptr(buf), SYNC_FLOATS),
                   "net_set_sync_bn")
        self._sync = state

    def _release_sync(self):
        s, self._sync = self._sync, None
        if s and s.get("comm"):
            _lib.check(_lib.get().lbc_net_set_sync_bn(self.handle, None, None, 1, None, 0), "net_set_sync_bn")
            _lib.get().lbc_comm_destroy(s["comm"])

    @staticmethod
    def num_stages():
        return _lib.get().lbc_net_num_stages()
