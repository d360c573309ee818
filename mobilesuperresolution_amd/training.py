"""Drop-ins for the two torch objects the reference trainers build next to the model, so that their training loops run
UNCHANGED on the fast route:

    pretrain.py:137      optim.Adam(filter(...model.parameters()), lr)        ->  training.Adam(...)
    pretrain.py:220      criterions = {'l1': nn.L1Loss()}                      ->  {'l1': training.L1Loss()}
    train_video_superresolution.py:43-53   L1_Charbonnier_loss()              ->  training.L1_Charbonnier_loss()

and then, as before (pretrain.py:69-82):

    optimizer.zero_grad(); sr = model(lr); loss = w * criterions['l1'](sr, hr); loss.backward(); optimizer.step(); loss.item()

What changes underneath.  `criterion(sr, hr)` recognises an `sr` that came straight out of BASIC_MODEL.forward (its grad_fn is
the network's autograd node; likewise NAS_MODEL's `sr, speed = model(lr)`, search.py:74: there the tail's node) and runs the network's backward right there with the loss folded into the tail-backward kernel
(csrc/wdsr_ends.h, sr_tail_bwd_loss): no d(loss)/d(sr) tensor, none of the seven small ATen loss kernels; `loss.backward()` then
only hands the finished parameter gradient over (scaled by whatever the trainer multiplied the loss with).  `Adam.step()` is ONE
launch of the library's Adam kernel per parameter tensor (BASIC_MODEL has one flat parameter) with torch.optim.Adam's
arithmetic bit for bit (tests/test_gpu_train_step.py), where torch's multi-tensor path takes ~43 us for it.  Any other tensor
(`sr` from another module, a sliced `sr`, no grad mode) takes torch's own ops: that is the reference's behaviour, not a fallback
of the hot path."""
from __future__ import annotations

import ctypes
import math

import torch
import torch.nn as nn

from . import _lib as L

__all__ = ["L1Loss", "L1_Charbonnier_loss", "Adam"]


class _BasicFold:
    """BASIC_MODEL: the node is the whole network's; folding runs the network's backward, the payload is the flat gradient"""

    @staticmethod
    def can_fold(node, sr, hr):
        return node.model._can_fold(node, sr, hr)

    @staticmethod
    def run(node, sr, hr, kind):
        model = node.model
        gflat, _keep = model._backward_folded(node, sr, hr, kind)
        st = model._state(sr.device)
        return gflat, st.loss_part


class _NodeFold:
    """NAS_MODEL: the node is the tail's (models/wdsr_b.py:_TailFunction); folding runs the tail's backward, the body's runs
    later under autograd"""

    @staticmethod
    def can_fold(node, sr, hr):
        return node.can_fold(node, sr, hr)

    @staticmethod
    def run(node, sr, hr, kind):
        return node.fold_loss(node, sr, hr, kind)


def _network_node(sr: torch.Tensor):
    """(autograd node that produced `sr`, its folding adapter) if it is one of this package's output nodes, else (None, None)"""
    fn = sr.grad_fn
    if fn is None:
        return None, None
    name = type(fn).__name__
    if name == "_NetFunctionBackward" and hasattr(fn, "model"):
        return fn, _BasicFold
    if name == "_TailFunctionBackward" and hasattr(fn, "fold_loss"):
        return fn, _NodeFold
    return None, None


class _FoldedCriterion(torch.autograd.Function):
    """loss(sr, hr) where sr = BASIC_MODEL(x) or NAS_MODEL(x)[0]: forward() runs the BACKWARD of sr's node with the loss gradient
    formed inside the tail-backward kernel and returns the loss value; backward() parks the result on the node and sends a
    zero-stride zero token up the graph in place of d(loss)/d(sr)."""

    @staticmethod
    def forward(ctx, sr, hr, node, fold, kind):
        payload, loss_part = fold.run(node, sr, hr, kind)
        loss = torch.empty((), dtype=torch.float32, device=sr.device)
        with L.device_guard(sr.device):
            L.launch("sr_loss_value", L.lib().sr_loss_value, loss_part.data_ptr(), loss_part.numel(), 1.0 / sr.numel(), loss.data_ptr(),
                     L.stream_ptr(sr.device))
        ctx.node = node
        ctx.token = torch.zeros(1, dtype=sr.dtype, device=sr.device).expand(sr.shape)
        ctx.payload = payload
        return loss

    @staticmethod
    def backward(ctx, gloss):
        ctx.node.folded = (ctx.payload, gloss, ctx.token.data_ptr())
        return ctx.token, None, None, None, None


class _HotLoss(nn.Module):
    KIND = "l1"

    def forward(self, sr: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
        node, fold = _network_node(sr) if (torch.is_grad_enabled() and sr.requires_grad) else (None, None)
        if node is None or getattr(node, "folded", None) is not None or not fold.can_fold(node, sr, hr):
            return self._torch_loss(sr, hr)
        return _FoldedCriterion.apply(sr, hr.detach().contiguous().float(), node, fold, self.KIND)


class L1Loss(_HotLoss):
    """nn.L1Loss() (mean reduction) -- pretrain.py:220 / :73"""
    KIND = "l1"

    def _torch_loss(self, sr, hr):
        return torch.nn.functional.l1_loss(sr, hr)


class L1_Charbonnier_loss(_HotLoss):
    """train_video_superresolution.py:43-53: mean(sqrt((X - Y)^2 + 1e-12))"""
    KIND = "charbonnier"

    def __init__(self):
        super().__init__()
        self.eps = 1e-12

    def _torch_loss(self, sr, hr):
        diff = torch.add(sr, -hr)
        return torch.mean(torch.sqrt(diff * diff + self.eps))


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) with the step of every parameter tensor as ONE launch of csrc/train_step.h's
    adam_step_kernel (same fused-multiply-add forms as torch's foreach implementation: parameters bit-identical).  State keys
    are torch's (`step`, `exp_avg`, `exp_avg_sq`), so optimizer checkpoints interchange (pretrain.py:262-267); schedulers
    (MultiStepLR, pretrain.py:139-142) act on `param_groups[i]['lr']` as usual.  weight_decay / amsgrad / maximize: not on the
    reference's path, rejected."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("training.Adam mirrors the reference's optim.Adam(params, lr): no weight decay, no amsgrad")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad's effect without its per-call bookkeeping (profiler record, foreach grouping: ~12 us of host
        time for one parameter -- in the reference's loop that sits between the per-step `loss.item()` sync and the forward's first
        launch, i.e. it is GPU idle time)"""
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_()
                        p.grad.requires_grad_(False)
                        p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        for group in self.param_groups:
            lr, (b1, b2), eps = float(group["lr"]), group["betas"], float(group["eps"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or g.dtype != torch.float32 or g.is_sparse:
                    raise L.HotpathError("training.Adam steps contiguous fp32 CUDA parameters (there is no CPU fallback)")
                g = g.contiguous()
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                t = int(state["step"])
                bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
                scal = L.AdamScalars(1.0 - b1, b2, 1.0 - b2, math.sqrt(bc2), eps, -(lr / bc1))
                with L.device_guard(p.device):
                    L.launch("sr_adam_step", lib.sr_adam_step, p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(),
                             state["exp_avg_sq"].data_ptr(), p.numel(), ctypes.byref(scal), None, 0, 0.0, None, L.stream_ptr(p.device))
                try:
                    torch.autograd.graph.increment_version(p)
                except AttributeError:
                    p.add_(0)
        return loss
