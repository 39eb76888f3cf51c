"""ctypes binding of libstgcn_hip.so (the C-ABI library of gfx950 kernels).

There is no fallback: if the library is missing or a call returns a non-zero status,
this module raises.  Signatures mirror include/stgcn_hip.h.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("STGCN_LIB") or os.path.join(_HERE, "libstgcn_hip.so")     # STGCN_LIB: another build (A/B runs)

F32, BF16 = 0, 1
EPI_PLAIN, EPI_RELU_BWD, EPI_ADD_RELU = 0, 1, 2

_lib = None

_vp, _i, _f, _d, _sz, _u32 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint32

_SIGS = {
    "stgcn_conv_rows": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                        _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stgcn_conv_rows_bn": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _f, _f, _vp,
                           _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "stgcn_conv_wgrad": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp],
    "stgcn_gcn_fused": [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stgcn_gcn_agrad": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp],
    "stgcn_gcn_agrad_noalloc": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stgcn_gcn_wgrad": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp],
    "stgcn_conv_wgrad_deferred": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp],
    "stgcn_gcn_agrad_deferred": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp],
    "stgcn_gcn_wgrad_deferred": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp],
    "stgcn_sum_list_add_cast": [_vp, _vp, _vp, _sz],
    "stgcn_sum_list_run": [_vp, _vp],
    "stgcn_bn_finalize": [_vp, _d, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "stgcn_stats_to_grads": [_vp, _vp, _vp, _i, _vp],
    "stgcn_block_out_fwd": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _f, _u32, _vp, _vp],
    "stgcn_block_out_fwd_bn": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f,
                               _sz, _i, _f, _u32, _vp, _vp],
    "stgcn_block_out_bwd": [_i, _i, _vp, _vp, _sz, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _f, _u32, _d, _i, _vp],
    "stgcn_bn_bwd_apply": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _sz, _i, _i, _vp],
    "stgcn_augment": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stgcn_data_bn_stats": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "stgcn_data_bn_apply": [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stgcn_data_bn_bwd": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "stgcn_pool_fwd": [_i, _vp, _vp, _i, _i, _i, _vp],
    "stgcn_head_fc_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "stgcn_head_fc_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "stgcn_pool_bwd": [_i, _vp, _vp, _i, _i, _i, _i, _vp],
    "stgcn_mul_many": [_vp, _vp, _i, _vp, _sz, _vp],
    "stgcn_cross_entropy": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "stgcn_pack_weights": [_i, _vp, _i, _sz, _vp],
    "stgcn_sgd_step": [_vp, _vp, _vp, _sz, _f, _f, _f, _i, _i, _f, _vp],
    "stgcn_probe_tr16": [_vp, _vp],
    "stgcn_version": [],
    "stgcn_device_count": [],
    "stgcn_launch_state": [_i, _vp, _vp],
}

EXPORTS = tuple(_SIGS) + ("stgcn_arch",)
_ERR = {1: "invalid argument", 2: "kernel launch failed", 3: "workspace too small",
        4: "the problem exceeds the library's static scratch array", 5: "the sum list is full"}
SUM_LIST_BYTES = 1024


def lib_path():
    return _LIB_PATH


def load():
    """Loads the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            "libstgcn_hip.so is missing (%s): build it with `make -C st-gcn_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback" % _LIB_PATH)
    lib = C.CDLL(_LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _i
    lib.stgcn_arch.restype = C.c_char_p
    lib.stgcn_arch.argtypes = []
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The current stream's handle.  torch.cuda.current_stream() builds a Stream object per call (about 9 us: a
    fifth of the host's time per training step, which is what bounds small batches); the raw accessor does not."""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError("stgcn kernels take float32 or bfloat16 activations, got %s" % dt)


_SYNC = os.environ.get("STGCN_SYNC", "0") == "1"     # debugging aid: synchronise after every launch


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if _SYNC:
        torch.cuda.synchronize()
    if rc != 0:
        raise RuntimeError("%s failed: %s (status %d)" % (name, _ERR.get(rc, "?"), rc))
