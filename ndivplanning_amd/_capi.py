"""ctypes binding of libndp_hip.so (include/ndp.h) + tensor-level call helpers.

This is the only place that touches the shared library.  Loading fails loudly
(ImportError-like RuntimeError) when the library is missing: the product has no
CPU or PyTorch-eager fallback.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

import torch

from . import _build

CODE_DIM = 256
ACTION_DIM = 4
MAX_NOISE_DIM = 16
MAX_SAMPLES = 256
ROW_PAD = 32
EXPECTED_VERSION = 135          # NDP_VERSION of include/ndp.h this binding was written against

_lib = None


class NdpError(RuntimeError):
    pass


P2P_MAX_RANKS = 8
P2P_HANDLE_BYTES = 64
P2P_DIAG_WORDS = 8


class P2P(Structure):
    """struct ndp_p2p (include/ndp.h)."""
    _fields_ = [("world", c_int32), ("rank", c_int32), ("timeout_ms", c_int32), ("reserved", c_int32),
                ("region", c_void_p * P2P_MAX_RANKS)]


class StepConfig(Structure):
    """struct ndp_step_config (include/ndp.h)."""
    _fields_ = [("noise_dim", c_int32), ("num_sample", c_int32), ("flat", c_int64),
                ("inv_m_global", c_float), ("pairwise_div_factor", c_float),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("fuse_adam", c_int32), ("device_noise", c_int32), ("noise_seed", c_uint64),
                ("p2p", POINTER(P2P))]


class StepBuffers(Structure):
    """struct ndp_step_buffers (include/ndp.h)."""
    _fields_ = [(n, c_void_p) for n in (
        "g_params", "g_grad", "g_exp_avg", "g_exp_avg_sq",
        "d_params", "d_grad", "d_exp_avg", "d_exp_avg_sq",
        "g_step", "d_step", "losses", "loss_sums", "action_hat", "workspace")]


# name -> (restype, argtypes); the symbol list tests/test_capi_symbols.py checks against ndp.h
SIGNATURES = {
    "ndp_version": (c_int, []),
    "ndp_last_error": (c_char_p, []),
    "ndp_g_param_count": (c_int64, [c_int]),
    "ndp_d_param_count": (c_int64, []),
    "ndp_pad_rows": (c_int64, [c_int64]),
    "ndp_ndiv_partials": (c_int64, [c_int64, c_int]),
    "ndp_ndiv_fwd_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_float,
                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "ndp_g_acts_floats": (c_int64, [c_int64]),
    "ndp_g_forward": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64,
                              c_void_p, c_void_p, c_void_p]),
    "ndp_g_bwd_ws_floats": (c_int64, [c_int64, c_int]),
    "ndp_g_backward": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ndp_d_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "ndp_d_bwd_ws_floats": (c_int64, [c_int64]),
    "ndp_d_backward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int64, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "ndp_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                              c_float, c_float, c_float, c_float, c_void_p]),
    "ndp_step_workspace_floats": (c_int64, [POINTER(StepConfig)]),
    "ndp_step_d_grads": (c_int, [POINTER(StepConfig), POINTER(StepBuffers), c_void_p, c_void_p, c_void_p,
                                 c_int, c_void_p]),
    "ndp_step_g_grads": (c_int, [POINTER(StepConfig), POINTER(StepBuffers), c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    "ndp_step_pack_params": (c_int, [POINTER(StepConfig), POINTER(StepBuffers), c_void_p]),
    "ndp_step_apply_adam": (c_int, [POINTER(StepConfig), POINTER(StepBuffers), c_int, c_void_p]),
    "ndp_uniform_noise": (c_int, [c_void_p, c_int64, c_uint64, c_void_p, c_void_p]),
    "ndp_p2p_region_bytes": (c_int64, []),
    "ndp_p2p_region_alloc": (c_int, [POINTER(c_void_p)]),
    "ndp_p2p_region_free": (c_int, [c_void_p]),
    "ndp_p2p_region_reset": (c_int, [c_void_p]),
    "ndp_p2p_export": (c_int, [c_void_p, c_void_p]),
    "ndp_p2p_open": (c_int, [c_void_p, POINTER(c_void_p)]),
    "ndp_p2p_close": (c_int, [c_void_p]),
    "ndp_p2p_status": (c_int, [POINTER(P2P), POINTER(c_int32)]),
    "ndp_p2p_diagnostics": (c_int, [POINTER(P2P), POINTER(c_int32)]),
    "ndp_p2p_status_async": (c_int, [POINTER(P2P), c_void_p, c_void_p]),
    "ndp_device_pci_bus_id": (c_int, [ctypes.c_char_p, c_int]),
    "ndp_p2p_all_reduce": (c_int, [POINTER(P2P), c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "ndp_encoder_param_floats": (c_int64, []),
    "ndp_encoder_workspace_floats": (c_int64, [c_int64]),
    "ndp_encoder_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "ndp_encoder_forward_u8": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "ndp_fm_param_floats": (c_int64, []),
    "ndp_fm_stat_floats": (c_int64, []),
    "ndp_fm_workspace_floats": (c_int64, [c_int64]),
    "ndp_fm_workspace_offset": (c_int64, [c_int64, c_int]),
    "ndp_fm_layout": (c_int, [c_int, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "ndp_fm_pack_params": (c_int, [c_void_p, c_void_p, c_void_p]),
    "ndp_fm_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "ndp_fm_train_grads": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "ndp_fm_forward_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "ndp_fm_train_grads_u8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    "ndp_fm_side_stream": (c_int, [c_int]),
    "ndp_fm_grad_buckets": (c_int, [POINTER(c_int64), POINTER(c_int64), c_int]),
    "ndp_fm_set_stat_sync": (c_int, [c_void_p, c_void_p, c_int]),
    "ndp_fm_bucket_wait": (c_int, [c_int, c_void_p]),
    "ndp_fm_backward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "ndp_fm_apply_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                                  c_void_p, c_void_p]),
    "ndp_timing_enable": (c_int, [c_int]),
    "ndp_timing_collect": (c_int, [ctypes.c_char_p, c_int, POINTER(c_float), POINTER(c_int32), c_int]),
}


def lib_path():
    return _build.LIB_PATH


def load():
    """Load libndp_hip.so and declare every prototype.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise NdpError(
            "ndivplanning_amd: %s is missing -- build it with `python -m ndivplanning_amd._build` "
            "(or __graft_entry__.build()); there is no fallback path." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ndp_version() != EXPECTED_VERSION:
        raise NdpError("%s is version %d, this package needs %d: rebuild with `python -m ndivplanning_amd._build`"
                       % (path, lib.ndp_version(), EXPECTED_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ndp_last_error()
        raise NdpError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def stream_ptr(device=None):
    """The current HIP stream of `device` (default: the current device).  Launches must run on the
    device that owns their buffers: take this inside `on_device(...)`, or pass the device."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device(t):
    """Context manager that makes the device of tensor / torch.device `t` current for the launches
    inside it.  The library launches on whatever device is current (kernel attributes, stream), so a
    module that lives on cuda:1 while cuda:0 is current -- the reference's default `gpu_id: 1` with
    `torch.load(...).to(gpu_id)` and no set_device -- must switch for the duration of the call."""
    dev = t.device if isinstance(t, torch.Tensor) else torch.device(t)
    return torch.cuda.device(dev)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def require_gpu_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise NdpError("%s is on %s: ndivplanning_amd computes only on a ROCm GPU (no CPU fallback)"
                       % (name, t.device))
    if t.dtype != torch.float32:
        raise NdpError("%s must be float32, got %s" % (name, t.dtype))
    return t


def empty(n, like):
    return torch.empty(int(n), dtype=torch.float32, device=like.device)


def timing_enable(on):
    load().ndp_timing_enable(1 if on else 0)


def timing_collect():
    """{kernel name: (total ms, launches)} since timing was enabled / last collected."""
    lib = load()
    names = ctypes.create_string_buffer(4096)
    ms = (c_float * 64)()
    counts = (c_int32 * 64)()
    n = lib.ndp_timing_collect(names, 4096, ms, counts, 64)
    keys = [k for k in names.value.decode().split(";") if k]
    return {keys[i]: (float(ms[i]), int(counts[i])) for i in range(min(n, len(keys)))}


def g_param_count(noise_dim):
    return int(load().ndp_g_param_count(int(noise_dim)))


def d_param_count():
    return int(load().ndp_d_param_count())


def pad_rows(m):
    return (int(m) + ROW_PAD - 1) // ROW_PAD * ROW_PAD


def fm_grad_buckets():
    """[(offset, count)]: the ranges of the forward model's flat gradient in completion order (ndp_fm_grad_buckets)."""
    lib = load()
    off = (ctypes.c_int64 * 16)()
    cnt = (ctypes.c_int64 * 16)()
    n = lib.ndp_fm_grad_buckets(off, cnt, 16)
    if n <= 0:
        msg = lib.ndp_last_error()
        raise NdpError("ndp_fm_grad_buckets failed: %s" % (msg.decode() if msg else "?"))
    return [(int(off[i]), int(cnt[i])) for i in range(n)]


STAT_SYNC_FN = ctypes.CFUNCTYPE(None, c_void_p, c_int64, c_void_p, c_void_p)    # ndp_fm_stat_sync_fn (include/ndp.h)
