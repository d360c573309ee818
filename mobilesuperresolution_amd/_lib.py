"""ctypes binding of libsr_hotpath.so (include/sr_hotpath.h).  There is no CPU or ATen fallback:
if the library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SR_HOTPATH_DEBUG_LIB=1 (tools/ only): the diagnostic build with in-kernel time stamps (build.py --debug);
# SR_HOTPATH_LIB_PATH (tools/ only): an explicitly named build of the same sources (A/B timing of kernel variants)
LIB_PATH = os.environ.get("SR_HOTPATH_LIB_PATH") or os.path.join(
    _HERE, "libsr_hotpath_dbg.so" if os.environ.get("SR_HOTPATH_DEBUG_LIB") == "1" else "libsr_hotpath.so")
ABI_VERSION = 13
DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}

_P, _I, _Z, _L, _F = c_void_p, c_int, c_size_t, ctypes.c_long, ctypes.c_float
# name -> (argtypes, restype); must list every symbol include/sr_hotpath.h declares
SIGNATURES = {
    "sr_abi_version": ([], _I),
    "sr_wdsr_block_fwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "sr_wdsr_block2_fwd": ([_P] * 9 + [_I] * 5 + [_P], _I),
    "sr_wdsr_fwd_rs": ([_P] * 9 + [_I] * 6 + [_P], _I),
    "sr_conv7_fwd": ([_P] * 4 + [_I] * 7 + [_P], _I),
    "sr_wdsr_fwd_rs_repeat": ([_P] * 9 + [_I] * 7 + [_P], _I),
    "sr_wdsr_block_wgrad_saved": ([_P] * 8 + [_I] * 7 + [_L] * 5 + [_P], _I),
    "sr_wdsr_block2_bwd_data": ([_P] * 11 + [_I] * 5 + [_P], _I),
    "sr_wdsr_block2_fwd_repeat": ([_P] * 7 + [_I] * 6 + [_P], _I),
    "sr_probe_launch_floor": ([_P, _I, _I, _I, _I, _I, _P], _I),
    "sr_probe_launch_floor_graph": ([_P, _I, _I, _I, _I, _I, _I, _P, _P], _I),
    "sr_debug_set_stamps": ([_P, _L], _I),
    "sr_c3_trunk_fwd": ([_P] * 6 + [_I] * 7 + [_L, _P], _I),
    "sr_c3_trunk_bwd": ([_P] * 11 + [_I] * 8 + [_L, _P], _I),
    "sr_tail_bwd": ([_P, _P, _P, _F, _P, _P, _P] + [_I] * 7 + [_P], _I),
    "sr_nas_dw_wgrad": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_wdsr_block_fwd_repeat": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_wdsr_block_bwd_data": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "sr_wdsr_block_wgrad": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _L, _P], _I),
    "sr_wdsr_block_slab_sizes": ([_I, _P, _P], _I),
    "sr_head_fwd": ([_P, _P, _P, _F, _I, _I, _I, _I, _I, _P], _I),
    "sr_tail_fwd": ([_P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_tail_bwd_data": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_tail_wgrad": ([_P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_head_wgrad": ([_P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_c3_fwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_c3_bwd_data": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_c3_wgrad": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_flow_warp_fwd": ([_P, _P, _P, _I, _I, _I, _I, _P], _I),
    "sr_flow_warp_bwd": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _P], _I),
    "sr_nas_dw_fwd": ([_P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "sr_nas_pw_fwd": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "sr_nas_pw_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_nas_dw_bwd": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_patch_gather": ([_P, _P, _P, _P, _I, _I, _I, _P], _I),
    "sr_param_pack": ([_P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _P], _I),
    "sr_param_grads": ([_P, _P, _P, _P, _I, _P, _I, _P, _I, _P], _I),
    "sr_nas_scalars": ([_P] * 5 + [_I, _I, _P, _P, _L, _I, _P, _P], _I),
    "sr_nas_mask_grads": ([_P, _L, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P], _I),
    "sr_nas_body_fwd": ([_P, _P, _P, _L, _P, _L, _P, _L, _P, _L] + [_I] * 6 + [_P], _I),
    "sr_nas_body_bwd": ([_P] * 7 + [_L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L] + [_I] * 7 + [_P, _P], _I),
    "sr_psnr": ([_P, _P, _P, _P] + [_I] * 7 + [_P], _I),
    "sr_pixel_shuffle": ([_P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_tail_bwd_loss": ([_P, _P, _I, _F, _P, _P, _P, _F, _P, _P, _P] + [_I] * 7 + [_P], _I),
    "sr_adam_step": ([_P, _P, _P, _P, _L, _P, _P, _I, _F, _P, _P], _I),
    "sr_loss_value": ([_P, _I, _F, _P, _P], _I),
    "sr_scale_by": ([_P, _P, _L, _P, _I, _P], _I),
    "sr_wdsr_net_train_step": ([_P, _P, _P, _L, _P, _F, _P, _P], _I),
    "sr_wdsr_net_forward": ([_P, _I, _P], _I),
    "sr_wdsr_net_backward": ([_P, _P], _I),
    "sr_wdsr_net_backward_part": ([_P, _I, _P], _I),
    "sr_probe_mfma_bf16": ([_P, _P, _P, _P], _I),
    "sr_probe_mfma_f32": ([_P, _P, _P, _P], _I),
    "sr_probe_tr_read": ([_P, _I, _P, _P, _P], _I),
    "sr_probe_copy": ([_P, _P, _Z, _P], _I),
}

class WdsrNet(ctypes.Structure):
    """mirror of sr_wdsr_net_t (include/sr_hotpath.h); field order and types must match"""
    _fields_ = (
        [(n, _I) for n in ("F", "NB", "R", "dtype", "N", "H", "W")] + [("mean", _F)] +
        [("flat", _P), ("gflat", _P), ("chan_tab", _P), ("n_chan", _I),
         ("bias_tab", _P), ("bias_const", _P), ("n_bias", _I), ("src", _P), ("dsrc", _P)] +
        [(n, _L) for n in ("src_head_off", "src_body_off", "src_body_stride", "src_tail_off")] +
        [("idx_head", _P), ("n_idx_head", _I), ("idx_body", _P), ("n_idx_body", _I),
         ("idx_cinit", _P), ("n_idx_cinit", _I), ("idx_tail", _P), ("n_idx_tail", _I),
         ("blob_head", _P), ("blob_body", _P), ("cinit_body", _P), ("blob_tail", _P),
         ("part_a", _P), ("part_b", _P), ("part_tail", _P), ("part_head", _P)] +
        [(n, _I) for n in ("wgs_body", "wgs_tail", "wgs_head", "slab_a", "slab_b", "slab_tail", "slab_head")] +
        [("ga_sidx", _P), ("ga_dst", _P), ("n_ga", _I), ("gb_sidx", _P), ("gb_dst", _P), ("n_gb", _I),
         ("gt_sidx", _P), ("gt_dst", _P), ("n_gt", _I), ("gh_sidx", _P), ("gh_dst", _P), ("n_gh", _I),
         ("x", _P), ("acts", _P), ("grads", _P), ("out", _P), ("dout", _P), ("tsave", _P), ("dtsave", _P),
         ("hr", _P), ("loss_kind", _I), ("loss_gscale", _F), ("loss_part", _P),
         ("nb_split", _I), ("chan_split", _I), ("bias_split", _I), ("adam_in_wn_bwd", _I)])


class C3Warp(ctypes.Structure):
    """mirror of sr_c3_warp_t"""
    _fields_ = [("frame", _P), ("frame_bs", _L), ("state", _P), ("flow", _P), ("flow_bs", _L), ("flow_bound", _P),
                ("dstate", _P), ("dflow", _P), ("dflow_bs", _L), ("x0_save", _P)]


class C3Unpack(ctypes.Structure):
    """mirror of sr_c3_unpack_t"""
    _fields_ = [("sidx0", _P), ("dst0", _P), ("n0", _I), ("sidx1", _P), ("dst1", _P), ("n1", _I), ("gflat", _P)]


class PackSeg(ctypes.Structure):
    """mirror of sr_pack_seg_t"""
    _fields_ = [("idx", _P), ("out", _P), ("src_off", _L), ("src_stride", _L), ("n", _I), ("reps", _I), ("as_float", _I)]


class UnpackSeg(ctypes.Structure):
    """mirror of sr_unpack_seg_t"""
    _fields_ = [("partial", _P), ("sidx", _P), ("dst", _P), ("dst_off", _L), ("dst_stride", _L), ("slab", _L), ("wgs", _I),
                ("n", _I), ("reps", _I)]


class AdamScalars(ctypes.Structure):
    """mirror of sr_adam_t"""
    _fields_ = [(n, _F) for n in ("w_lerp", "beta2", "one_minus_beta2", "bc2_sqrt", "eps", "neg_step_size")]


_lib = None


class HotpathError(RuntimeError):
    pass


def _declare(lib):
    for name, (args, res) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
        fn.argtypes = args
        fn.restype = res


def lib():
    """Load the shared library once; never builds, never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HotpathError(
                f"{LIB_PATH} not found: build it with `python -m mobilesuperresolution_amd.build` "
                "(the SR hot path has no CPU / ATen fallback)")
        if os.environ.get("SR_HOTPATH_LIB_PATH") or os.environ.get("SR_HOTPATH_DEBUG_LIB") == "1":
            import warnings
            warnings.warn(f"SR hot path: loading a NON-DEFAULT library ({LIB_PATH}) because SR_HOTPATH_LIB_PATH / "
                          "SR_HOTPATH_DEBUG_LIB is set (meant for tools/ only)", RuntimeWarning, stacklevel=2)
        l = ctypes.CDLL(LIB_PATH)
        _declare(l)
        if l.sr_abi_version() != ABI_VERSION:
            raise HotpathError(f"libsr_hotpath ABI {l.sr_abi_version()} != expected {ABI_VERSION}; rebuild")
        _lib = l
    return _lib


def stream_ptr(device=None) -> int:
    """the current HIP stream of `device` (default: the current device)"""
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    """`with` target that does nothing"""
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NOGUARD = _NoGuard()


def device_guard(device):
    """`with torch.cuda.device(device)` only when `device` is not the current one already: the context manager costs ~10 us of host time
    per use, and the training loop of the reference enters four of them per step between its per-step `loss.item()` syncs"""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NOGUARD
    return torch.cuda.device(idx)


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "unsupported geometry/dtype", -2: "bad argument"}.get(rc, f"hipError_t {rc}")
        raise HotpathError(f"{what} failed: {kind}")


def ptr(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise HotpathError("SR hot path needs device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise HotpathError("SR hot path needs contiguous tensors")
    return t.data_ptr()


class KernelTimer:
    """Optional per-call timing with HIP events on the launch stream (bench.py's roofline leg).
    Off by default: the product path records nothing."""

    def __init__(self):
        self.events = {}

    def run(self, name, fn, *args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        self.events.setdefault(name, []).append((e0, e1))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.events.items()}


_TIMER = None


def set_timer(t):
    global _TIMER
    _TIMER = t


def launch(name, fn, *args):
    rc = fn(*args) if _TIMER is None else _TIMER.run(name, fn, *args)
    check(rc, name)
