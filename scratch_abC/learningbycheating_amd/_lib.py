"""ctypes binding of the C ABI in include/lbc_hip.h (learningbycheating_amd/liblbc_hip.so).

The product path has NO fallback: if the gfx950 library is missing or a kernel launch
fails, a RuntimeError is raised.  (tests/emu injects a CPU-emulated build of the same
kernel sources through _inject_for_tests(); nothing in the package does.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblbc_hip.so")
_lib = None

c_void_p, c_int, c_float, c_size_t, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_char_p


ABI_VERSION = 200     # include/lbc_hip.h LBC_HIP_ABI_VERSION: load() refuses a library that answers anything else


class ConvDesc(ctypes.Structure):
    """lbc_conv_desc; struct_size (its first member, checked by every entry point) is filled in here: ConvDesc(N, H, W, C, K, ...)"""
    _fields_ = ([("struct_size", ctypes.c_uint)] +
                [(n, c_int) for n in ("N", "H", "W", "C", "K", "KH", "KW", "S", "P", "relu", "bf16", "w_transposed")] +
                [("split_workspace", c_void_p), ("split_workspace_bytes", c_size_t)])

    def __init__(self, *args, **kw):
        super().__init__(ctypes.sizeof(ConvDesc), *args, **kw)


class NetDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("arch", "in_channels", "H", "W", "normalize", "max_batch", "precision")]


class Camera(ctypes.Structure):
    _fields_ = [(n, c_float) for n in ("w", "h", "fov", "world_y", "fixed_offset", "pixels_per_meter", "crop_size")]


class HeadDesc(ctypes.Structure):
    _fields_ = ([("h", c_void_p), ("N", c_int), ("OH", c_int), ("OW", c_int), ("act_bf16", c_int)] +
                [(n, c_void_p * 4) for n in ("mean", "invstd", "gamma", "beta", "w", "bias", "pos_x", "pos_y")] + [("cmd", c_void_p)])


class AugParams(ctypes.Structure):
    _fields_ = [("order", c_int * 8), ("n_ops", c_int), ("blur_pos", c_int), ("seed", ctypes.c_uint), ("blur_sigma", c_float),
                ("noise_scale", c_float), ("noise_pc", c_int), ("coarse_p", c_float), ("coarse_h", c_int), ("coarse_w", c_int),
                ("coarse_pc", c_int), ("dropout_p", c_float), ("dropout_pc", c_int), ("add", c_float * 3), ("multiply", c_float * 3),
                ("contrast", c_float * 3)]


class AdamChunk(ctypes.Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int), ("pad", c_int)]


_SIGNATURES = {
    "lbc_last_error": (c_char_p, []),
    "lbc_backend": (c_char_p, []),
    "lbc_version": (c_int, []),
    "lbc_conv2d_fwd": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 6 + [c_int, c_void_p, c_void_p, ctypes.POINTER(c_int), c_void_p]),
    "lbc_conv2d_dgrad": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 5),
    "lbc_weight_transpose_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "lbc_conv2d_wgrad_workspace": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "lbc_conv2d_wgrad": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 4 + [c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "lbc_deconv3x3s2_fwd": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 5 + [c_int, c_void_p, c_void_p, ctypes.POINTER(c_int), c_void_p]),
    "lbc_deconv3x3s2_dgrad": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 4),
    "lbc_conv2d_wgrad_group_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "lbc_conv2d_wgrad_group_workspace": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "lbc_conv2d_wgrad_group": (c_int, [ctypes.POINTER(ConvDesc), c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p]),
    "lbc_deconv3x3s2_wgrad_workspace": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "lbc_deconv3x3s2_wgrad": (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 4 + [c_int, c_void_p, c_float, c_void_p, c_void_p]),
    "lbc_net_create": (c_int, [ctypes.POINTER(NetDesc), ctypes.POINTER(c_void_p)]),
    "lbc_net_destroy": (None, [c_void_p]),
    "lbc_net_num_tensors": (c_int, [c_void_p]),
    "lbc_net_tensor_info": (c_int, [c_void_p, c_int, c_char_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "lbc_net_workspace_bytes": (c_size_t, [c_void_p]),
    "lbc_net_num_activations": (c_int, [c_void_p]),
    "lbc_net_activation_info": (c_int, [c_void_p, c_int, c_char_p, c_int, ctypes.POINTER(c_size_t), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "lbc_net_bind": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p)]),
    "lbc_net_forward": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 6),
    "lbc_net_forward_u8": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 6),
    "lbc_net_num_stages": (c_int, []),
    "lbc_net_last_forward": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_longlong)]),
    "lbc_net_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "lbc_net_set_frozen": (c_int, [c_void_p, c_int]),
    "lbc_net_set_sync_bn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "lbc_comm_unique_id": (c_int, [c_void_p]),
    "lbc_comm_create": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "lbc_comm_destroy": (None, [c_void_p]),
    "lbc_comm_world_size": (c_int, [c_void_p]),
    "lbc_comm_allreduce_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "lbc_loss": (c_int, [c_int, ctypes.POINTER(Camera), c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "lbc_phase2_weight": (c_int, [ctypes.POINTER(Camera), c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "lbc_adam_step": (c_int, [c_void_p, c_int] + [ctypes.c_double] * 5 + [c_int, c_void_p]),
    "lbc_bn_stats": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, ctypes.POINTER(c_int), c_void_p]),
    "lbc_bn_finalize_stats": (c_int, [c_void_p, c_int, c_int, ctypes.c_longlong] + [c_void_p] * 5 + [c_float, c_float, c_int] + [c_void_p] * 5),
    "lbc_bn_apply_relu_add_fwd": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_int] + [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "lbc_bn_bwd_workspace": (c_size_t, [c_int]),
    "lbc_bn_bwd": (c_int, [c_void_p] * 9 + [ctypes.c_longlong, c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p]),
    "lbc_maxpool3x3s2_fwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "lbc_maxpool3x3s2_bwd": (c_int, [c_void_p] * 9 + [ctypes.POINTER(c_int)] + [c_int] * 5 + [c_void_p]),
    "lbc_head_workspace": (c_size_t, [c_int]),
    "lbc_head_fwd": (c_int, [ctypes.POINTER(HeadDesc)] + [c_void_p] * 4),
    "lbc_head_bwd": (c_int, [ctypes.POINTER(HeadDesc)] + [c_void_p] * 4 + [ctypes.POINTER(c_void_p)] * 4 + [c_void_p, c_void_p]),
    "lbc_nchw_to_input": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "lbc_u8nhwc_to_input": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "lbc_stem_fwd": (c_int, [c_void_p] * 4 + [ctypes.POINTER(c_int)] + [c_int] * 5 + [c_void_p]),
    "lbc_stem_wgrad_workspace": (c_size_t, [c_int] * 4),
    "lbc_stem_wgrad": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "lbc_birdview_crop_u8": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "lbc_birdview_warp_crop_u8": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "lbc_augment_rgb_u8": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "lbc_config_set": (c_int, [c_char_p, ctypes.c_longlong]),
    "lbc_config_get": (ctypes.c_longlong, [c_char_p]),
    "lbc_profile_enable": (c_int, [c_int]),
    "lbc_profile_report": (c_int, [c_char_p, c_int]),
    "lbc_adam_profile_elems": (None, [ctypes.c_longlong]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _declare(lib):
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = the library does not export what include/lbc_hip.h declares
        fn.restype = res
        fn.argtypes = args
    v = lib.lbc_version()
    if v != ABI_VERSION:
        raise RuntimeError("learningbycheating_amd: the library answers ABI version %d, this binding was written for %d "
                           "(include/lbc_hip.h LBC_HIP_ABI_VERSION): rebuild liblbc_hip.so" % (v, ABI_VERSION))
    return lib


def load(path=None):
    """Load the gfx950 library (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            "learningbycheating_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    _lib = _declare(ctypes.CDLL(path))
    return _lib


def get():
    return _lib if _lib is not None else load()


def _inject_for_tests(lib):
    """tests/emu only: use a CPU-emulated build of the kernel sources."""
    global _lib
    _lib = _declare(lib) if lib is not None else None
    return _lib


def config_set(name, value):
    """runtime option of the library (names = the LBC_* environment variables); value -1 = unset"""
    check(get().lbc_config_set(name.encode(), int(value)), "config_set")


def config_get(name):
    return int(get().lbc_config_get(name.encode()))


def backend():
    return get().lbc_backend().decode()


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("lbc_hip %s failed (%d): %s" % (what, rc, get().lbc_last_error().decode()))


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_for(t):
    import torch
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def require_device(t):
    """Kernels run on a ROCm device only (the emulated test build accepts CPU tensors)."""
    if not t.is_cuda and backend() != "emu-cpu":
        raise RuntimeError("learningbycheating_amd: tensors must live on a ROCm (cuda) device; there is no CPU path")
