"""Host-side mirror of the reference's `models` package for the SR hot path
(reference: models/__init__.py:5-6,31-32).  `get_model(params)` resolves
`params.model_type` exactly as the reference's `eval(params.model_type)(params)` does."""
from __future__ import annotations

import argparse

from .basic_wdsr_b import BASIC_MODEL
from .basicvsr_arch import BasicVSR, ConvResidualBlocks, ResidualBlockNoBN
from .basicvsr_arch_origin import BasicVSR_origin, pixel_shuffle
from .mvvsr_arch import MotionVectorVSR
from .spynet_arch import SpyNet, flow_warp
from .wdsr_b import NAS_MODEL, ModelOutput

__all__ = ["BASIC_MODEL", "NAS_MODEL", "ModelOutput", "ConvResidualBlocks", "ResidualBlockNoBN", "MotionVectorVSR",
           "BasicVSR_origin", "BasicVSR", "SpyNet", "flow_warp", "pixel_shuffle", "get_model", "update_argparser", "wrap_ddp"]

_REGISTRY = {"BASIC_MODEL": BASIC_MODEL, "NAS_MODEL": NAS_MODEL}


def update_argparser(parser: argparse.ArgumentParser):
    """The model flags the trainers pass through (reference: models/__init__.py:9-29)."""
    parser.add_argument('--learning_rate', help='Learning rate.', default=0.001, type=float)
    parser.add_argument('--pretrained', action='store_true', default=False)
    parser.add_argument('--width_search', action='store_true', default=False, help='Width Search.')
    parser.add_argument('--length_search', action='store_true', default=False)
    parser.add_argument('--num_blocks', help='Number of residual blocks in networks.', default=16, type=int)
    parser.add_argument('--num_residual_units', help='Number of residual units in networks.', default=24, type=int)
    # registered by the reference and read by nobody on this path (models/__init__.py:18-28); kept so that the
    # trainers' command lines parse unchanged
    parser.add_argument('--seperate', help='seperate conv', default=False, type=int)
    parser.add_argument('--bottleneck_type', help='inverted_bottle', default='inverted_bottle', type=str)
    parser.add_argument('--clip_range', help='weight clip range.', default=None, type=float)
    parser.add_argument('--trainable_clip', action='store_true', default=False, help='trainable clip.')
    parser.add_argument('--clip_quantile_lb', help='weight clip quantile lower bound.', default=None, type=float)
    parser.add_argument('--clip_quantile_ub', help='weight clip quantile upper bound.', default=None, type=float)
    parser.add_argument('--clip_range_tail', help='weight clip range for tail layer.', default=None, type=float)
    parser.add_argument('--clip_range_skip', help='weight clip range for skip layer.', default=None, type=float)
    parser.add_argument('--clip_scale', help='weight clip scale factor for quantile clipping.', default=1.0, type=float)
    parser.add_argument('--hot_dtype', help='MI355X hot-path storage/compute type: fp32 (exact) or bf16.',
                        default=None, type=str)


def get_model(params):
    try:
        cls = _REGISTRY[params.model_type]
    except KeyError:
        raise NotImplementedError(
            f"model_type {params.model_type!r} is not on the MI355X hot path (have {sorted(_REGISTRY)})")
    return cls(params)


def wrap_ddp(model, **ddp_kwargs):
    """`DistributedDataParallel(model, ...)` as the trainers build it (pretrain.py:239, search.py:294,332,375), plus the
    two things the hot path needs that stock defaults do not give:
      * BASIC_MODEL with two gradient segments: a bucket cap that separates the segments, so that the late half's
        all-reduce runs under the early half's backward (bucket_cap_mb, gradient_as_bucket_view);
      * NAS_MODEL: nothing -- the reference's `beta`, `beta1`, `beta2` (Parameters that never receive a gradient, on which
        DDP without find_unused_parameters raises) are slices of the model's one flat body parameter here, with a zero
        gradient, so neither freezing nor find_unused_parameters=True (a graph walk and a host sync per iteration) is needed.
    Call it again after each phase change of search.py:329-333,372-376 (unwrap with `.module`, length_grad / mask_grad,
    wrap_ddp)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if hasattr(model, "freeze_gradless_parameters"):
        model.freeze_gradless_parameters()
    if isinstance(model, BASIC_MODEL) and getattr(model, "grad_segments", 1) == 2:
        ddp_kwargs.setdefault("bucket_cap_mb", model.ddp_bucket_cap_mb())
        ddp_kwargs.setdefault("gradient_as_bucket_view", True)
    return DDP(model, **ddp_kwargs)
