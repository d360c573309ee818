"""ctypes binding of libsr_hotpath.so (include/sr_hotpath.h).  There is no CPU or ATen fallback:
if the library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsr_hotpath.so")
ABI_VERSION = 1
DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}

_P, _I, _Z, _L, _F = c_void_p, c_int, c_size_t, ctypes.c_long, ctypes.c_float
# name -> (argtypes, restype); must list every symbol include/sr_hotpath.h declares
SIGNATURES = {
    "sr_abi_version": ([], _I),
    "sr_wdsr_block_fwd": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "sr_wdsr_block_bwd_data": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], _I),
    "sr_wdsr_block_wgrad": ([_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _L, _P], _I),
    "sr_wdsr_block_slab_sizes": ([_I, _P, _P], _I),
    "sr_head_fwd": ([_P, _P, _P, _F, _I, _I, _I, _I, _I, _P], _I),
    "sr_tail_fwd": ([_P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_tail_bwd_data": ([_P, _P, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_tail_wgrad": ([_P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_head_wgrad": ([_P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _P], _I),
    "sr_probe_mfma_bf16": ([_P, _P, _P, _P], _I),
    "sr_probe_mfma_f32": ([_P, _P, _P, _P], _I),
    "sr_probe_tr_read": ([_P, _I, _P, _P, _P], _I),
    "sr_probe_copy": ([_P, _P, _Z, _P], _I),
}

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
        l = ctypes.CDLL(LIB_PATH)
        _declare(l)
        if l.sr_abi_version() != ABI_VERSION:
            raise HotpathError(f"libsr_hotpath ABI {l.sr_abi_version()} != expected {ABI_VERSION}; rebuild")
        _lib = l
    return _lib


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


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
