"""
ctypes binding of libwavenet_amd.so (C ABI: include/wavenet_amd.h).

There is deliberately NO fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  Nothing here touches torch.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwavenet_amd.so")

MAX_TAPS = 8
MAX_CHANNELS = 1024

c_float_p = c_void_p  # device pointers travel as integers


class BlockShape(Structure):
    """wn_block_shape"""
    _fields_ = [("batch", c_int), ("length", c_int), ("in_channels", c_int), ("out_channels", c_int),
                ("skip_rows", c_int), ("kernel_width", c_int), ("dilation", c_int), ("causal", c_int),
                ("ld", c_int), ("halo", c_int)]


class BlockParams(Structure):
    """wn_block_params (parameters or their gradients, PyTorch layouts)"""
    _fields_ = [(n, c_float_p) for n in ("w_tanh", "b_tanh", "w_sigmoid", "b_sigmoid", "w_res", "b_res",
                                         "w_skip", "b_skip", "w_proj", "b_proj")]


MAX_STACK_GROUP = 32


class SkipSumShape(Structure):
    """wn_skipsum_shape"""
    _fields_ = [("batch", c_int), ("length", c_int), ("skip_rows", c_int), ("nblocks", c_int), ("ld", c_int),
                ("halo", c_int), ("channels", c_int * MAX_STACK_GROUP)]


class MemRange(Structure):
    """wn_mem_range"""
    _fields_ = [("base", c_void_p), ("bytes", c_size_t)]


class ConvShape(Structure):
    """wn_conv_shape"""
    _fields_ = [("batch", c_int), ("length", c_int), ("in_channels", c_int), ("out_channels", c_int),
                ("kernel_width", c_int), ("dilation", c_int), ("causal", c_int), ("ld", c_int), ("halo", c_int)]


# every symbol include/wavenet_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "wn_version": (c_int, []),
    "wn_strerror": (c_char_p, [c_int]),
    "wn_last_hip_error": (c_char_p, []),
    "wn_round_up": (c_int, [c_int, c_int]),
    "wn_autopad": (c_int, [c_int, c_int]),
    "wn_tap_offsets": (c_int, [c_int, c_int, c_int, POINTER(c_int)]),
    "wn_series_layout": (c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "wn_series_floats": (c_size_t, [c_int, c_int, c_int]),
    "wn_block_packed_bytes": (c_size_t, [POINTER(BlockShape)]),
    "wn_block_pack": (c_int, [POINTER(BlockShape), POINTER(BlockParams), c_void_p, c_void_p]),
    "wn_block_forward": (c_int, [POINTER(BlockShape), c_void_p, c_float_p, c_float_p, c_float_p, c_int,
                                 c_float_p, c_float_p, c_void_p]),
    "wn_skipsum_packed_bytes": (c_size_t, [POINTER(SkipSumShape)]),
    "wn_skipsum_pack": (c_int, [POINTER(SkipSumShape), POINTER(c_void_p), c_float_p, c_void_p, c_void_p]),
    "wn_skipsum_forward": (c_int, [POINTER(SkipSumShape), c_void_p, POINTER(c_void_p), c_float_p, c_int, c_void_p]),
    "wn_block_backward_data": (c_int, [POINTER(BlockShape), c_void_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                       c_float_p, c_float_p, c_float_p, c_void_p]),
    "wn_block_wgrad_workspace_bytes": (c_size_t, [POINTER(BlockShape)]),
    "wn_block_backward_weights": (c_int, [POINTER(BlockShape), c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                          c_float_p, POINTER(BlockParams), c_void_p, c_size_t, c_void_p]),
    "wn_conv_packed_bytes": (c_size_t, [POINTER(ConvShape)]),
    "wn_conv_pack": (c_int, [POINTER(ConvShape), c_float_p, c_float_p, c_void_p, c_void_p]),
    "wn_conv_forward": (c_int, [POINTER(ConvShape), c_void_p, c_float_p, c_float_p, c_void_p]),
    "wn_conv_backward_data": (c_int, [POINTER(ConvShape), c_void_p, c_float_p, c_float_p, c_void_p]),
    "wn_conv_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvShape)]),
    "wn_conv_backward_weights": (c_int, [POINTER(ConvShape), c_float_p, c_float_p, c_float_p, c_float_p,
                                         c_void_p, c_size_t, c_void_p]),
    "wn_hseries_layout": (c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "wn_hseries_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "wn_hseries_residual_scale": (c_float, []),
    "wn_hseries_load": (c_int, [c_int, c_float_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float_p, c_void_p,
                                c_void_p]),
    "wn_hblock_packed_bytes": (c_size_t, [POINTER(BlockShape), c_int]),
    "wn_hblock_pack": (c_int, [POINTER(BlockShape), c_int, POINTER(BlockParams), c_void_p, c_void_p]),
    "wn_hblock_pack_checked": (c_int, [POINTER(BlockShape), c_int, POINTER(BlockParams), c_void_p, c_void_p, c_void_p]),
    "wn_hblock_forward_is_fused": (c_int, [POINTER(BlockShape), c_int]),
    "wn_hblock_backward_pair_is_fused": (c_int, [POINTER(BlockShape), POINTER(BlockShape), c_int]),
    "wn_hblock_backward_pair": (c_int, [POINTER(BlockShape), c_void_p, POINTER(BlockShape), c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wn_hblock_backward_input": (c_int, [POINTER(BlockShape), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float_p, c_float_p,
                                         c_void_p, c_float, c_void_p, c_void_p]),
    "wn_hblock_forward": (c_int, [POINTER(BlockShape), c_int, c_void_p, c_void_p, c_void_p, c_float_p, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "wn_hskipsum_packed_bytes": (c_size_t, [POINTER(SkipSumShape), c_int]),
    "wn_hskipsum_pack": (c_int, [POINTER(SkipSumShape), c_int, POINTER(c_void_p), c_float_p, c_void_p, c_void_p]),
    "wn_hskipsum_forward": (c_int, [POINTER(SkipSumShape), c_int, c_void_p, POINTER(c_void_p), c_float_p, c_int, c_void_p]),
    "wn_hblock_backward_data": (c_int, [POINTER(BlockShape), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_float_p, c_float_p, c_void_p, c_void_p]),
    "wn_hblock_backward_data_masked": (c_int, [POINTER(BlockShape), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "wn_hfeature_forward": (c_int, [c_int, c_float_p, c_float_p, c_float_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                    c_float, c_void_p, c_void_p]),
    "wn_hfeature_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "wn_hfeature_backward_weights": (c_int, [c_int, c_float_p, c_void_p, c_float, c_float_p, c_float_p, c_int, c_int, c_int, c_int, c_int,
                                             c_int, c_float_p, c_void_p, c_size_t, c_void_p]),
    "wn_hseries_load_pooled": (c_int, [c_int, c_float_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float_p, c_void_p,
                                       c_void_p]),
    "wn_series_load_pooled": (c_int, [c_float_p, c_float_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "wn_grad_scale": (c_int, [c_float_p, c_longlong, c_float, c_float_p, c_void_p, c_void_p]),
    "wn_pool_backward": (c_int, [c_float_p, c_float_p, c_int, c_int, c_int, c_int, c_void_p]),
    "wn_hskipsum_forward_series": (c_int, [POINTER(SkipSumShape), c_int, c_void_p, POINTER(c_void_p), c_void_p, c_float, c_float, c_void_p,
                                           c_void_p]),
    "wn_hconv_forward_series": (c_int, [POINTER(ConvShape), c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "wn_hconv_backward_data_series": (c_int, [POINTER(ConvShape), c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                              c_void_p]),
    "wn_hblocks_wgrad_group_max": (c_int, [POINTER(BlockShape), c_int]),
    "wn_hblocks_wgrad_workspace_bytes": (c_size_t, [POINTER(BlockShape), c_int, c_int]),
    "wn_hblocks_backward_weights": (c_int, [POINTER(BlockShape), c_int, c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                            POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(BlockParams), c_float_p,
                                            c_void_p, c_size_t, c_void_p]),
    "wn_hstack_pack_table_bytes": (c_size_t, [c_int]),
    "wn_hstack_pack_table_build": (c_int, [POINTER(BlockShape), POINTER(BlockParams), c_int, c_int, c_int, POINTER(MemRange), c_int,
                                           c_void_p, c_size_t, POINTER(c_size_t), POINTER(c_size_t), POINTER(c_size_t),
                                           POINTER(c_int), POINTER(c_int)]),
    "wn_hstack_pack_run": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p]),
    "wn_hblock_wgrad_workspace_bytes": (c_size_t, [POINTER(BlockShape), c_int]),
    "wn_hblock_backward_weights": (c_int, [POINTER(BlockShape), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, POINTER(BlockParams), c_float_p, c_void_p, c_size_t, c_void_p]),
    "wn_hconv_packed_bytes": (c_size_t, [POINTER(ConvShape), c_int]),
    "wn_hconv_pack": (c_int, [POINTER(ConvShape), c_int, c_float_p, c_float_p, c_float, c_void_p, c_void_p]),
    "wn_hconv_forward": (c_int, [POINTER(ConvShape), c_int, c_void_p, c_void_p, c_float_p, c_void_p]),
    "wn_hconv_backward_data": (c_int, [POINTER(ConvShape), c_int, c_void_p, c_void_p, c_float_p, c_float_p, c_void_p]),
    "wn_hconv_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvShape), c_int]),
    "wn_hconv_backward_weights": (c_int, [POINTER(ConvShape), c_int, c_void_p, c_void_p, c_float, c_float_p, c_float_p, c_float_p,
                                          c_void_p, c_size_t, c_void_p]),
    "wn_embed_forward": (c_int, [c_void_p, c_float_p, c_float_p, c_float_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "wn_synth_workspace_bytes": (c_size_t, [c_int, c_int]),
    "wn_synth_bases": (c_int, [c_ulonglong, c_int, c_int, c_void_p, c_void_p]),
    "wn_synth_signal": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_ulonglong, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p, c_void_p]),
    "wn_synth_quantize": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wn_ctc_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "wn_ctc_loss": (c_int, [c_float_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float_p, c_float_p,
                            c_void_p, c_size_t, c_void_p, c_void_p]),
    "wn_nll_partials": (c_size_t, [c_int, c_int]),
    "wn_nll_forward": (c_int, [c_float_p, c_void_p, c_float_p, c_float_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "wn_nll_backward": (c_int, [c_float_p, c_void_p, c_float_p, c_float_p, c_float_p, c_int, c_int, c_int, c_void_p]),
    "wn_prof_enable": (c_int, [c_int]),
    "wn_prof_reset": (c_int, []),
    "wn_prof_collect": (c_int, []),
    "wn_prof_num_kernels": (c_int, []),
    "wn_prof_kernel_name": (c_char_p, [c_int]),
    "wn_prof_get": (c_int, [c_int, POINTER(c_double), POINTER(c_longlong), POINTER(c_double)]),
}

_lib = None


def load():
    """Load (once) and return the ctypes library with typed signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "wavenet_speech_amd: %s not found. Build it with `python __graft_entry__.py` or "
            "`make -C wavenet_speech_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.wn_version() < 300:
        raise RuntimeError("libwavenet_amd.so too old")
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        lib = load()
        msg = lib.wn_strerror(status).decode()
        hip = lib.wn_last_hip_error().decode()
        raise RuntimeError("%s failed: %s%s" % (what, msg, (" [" + hip + "]") if hip and status == -4 else ""))


def tap_offsets(k, d, causal):
    off = (c_int * k)()
    check(load().wn_tap_offsets(k, d, int(bool(causal)), off), "wn_tap_offsets")
    return list(off)


def series_layout(length, max_abs_offset):
    ld, halo = c_int(), c_int()
    check(load().wn_series_layout(length, max_abs_offset, ctypes.byref(ld), ctypes.byref(halo)), "wn_series_layout")
    return ld.value, halo.value


PRECISIONS = {"f32": 0, "f16x3": 1, "f16": 2, "bf16": 3}   # wn_precision


def hseries_layout(length, max_abs_offset):
    ld, halo = c_int(), c_int()
    check(load().wn_hseries_layout(length, max_abs_offset, ctypes.byref(ld), ctypes.byref(halo)), "wn_hseries_layout")
    return ld.value, halo.value


def profile_read():
    """{kernel class name: (total_ms, launches, flops)} after wn_prof_collect()."""
    lib = load()
    check(lib.wn_prof_collect(), "wn_prof_collect")
    out = {}
    for i in range(lib.wn_prof_num_kernels()):
        ms, n, fl = c_double(), c_longlong(), c_double()
        check(lib.wn_prof_get(i, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "wn_prof_get")
        out[lib.wn_prof_kernel_name(i).decode()] = (ms.value, n.value, fl.value)
    return out
