"""BASIC_MODEL on the MI355X hot path.

Same constructor fields, forward signature, `.scale` attribute and state_dict keys as the
reference's BASIC_MODEL (models/basic_wdsr_b.py:16-93): `head.*`, `body.{i}.body.{0,2,3}.*`,
`tail.*`, `skip.0.*`, each conv stored as `bias`, `weight_g`, `weight_v` (153 keys for 16 blocks).

MI355X-first design: every parameter lives in ONE flat fp32 buffer (`self.flat`, the only
nn.Parameter); `state_dict()` / `load_state_dict()` expose and accept the reference's 153 named
tensors as views into it, so checkpoints interchange with the reference (pretrain.py:222-223,260-267)
while the optimizer, autograd and DDP each see a single tensor.  `forward` is ONE call into
libsr_hotpath.so (weight-norm, fragment packing, head conv, N fused residual blocks, fused
tail + skip + PixelShuffle + mean) and `backward` is one more.  Nothing of the SR arithmetic runs in
PyTorch and there is no CPU / ATen fallback: CPU tensors, unsupported widths or a missing library raise.
"""
from __future__ import annotations

import ctypes
import math
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from ..layout import get_layout

__all__ = ["BASIC_MODEL"]

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


def _hot_dtype(params) -> torch.dtype:
    name = getattr(params, "hot_dtype", None) or os.environ.get("SR_HOT_DTYPE", "fp32")
    if isinstance(name, torch.dtype):
        return name
    return _DTYPES[str(name).lower()]


class _DeviceState:
    """Per-device tables and persistent work buffers of one model instance."""

    def __init__(self, model: "BASIC_MODEL", device: torch.device):
        lay = model.layout
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        f32 = lambda n: torch.empty(n, dtype=torch.float32, device=device)
        self.chan_tab, self.bias_tab = t(lay.chan_tab), t(lay.bias_tab)
        bc = lay.bias_const.copy()
        bc[np.isnan(bc)] = model.image_mean
        self.bias_const = t(bc)
        self.idx = {k: t(getattr(lay, "idx_" + k)) for k in ("head", "body", "cinit", "tail")}
        self.g = {k: (t(getattr(lay, k)[0]), t(getattr(lay, k)[1])) for k in ("ga", "gb", "gt", "gh")}
        self.src = torch.zeros(lay.src_total, dtype=torch.float32, device=device)
        self.src[torch.from_numpy(lay.src_ones).to(device)] = 1.0
        self.dsrc = torch.zeros(lay.src_total, dtype=torch.float32, device=device)
        dt = model.hot_dtype
        self.blob_head = torch.empty(lay.idx_head.size, dtype=dt, device=device)
        self.blob_body = torch.empty((lay.NB, lay.idx_body.size), dtype=dt, device=device)
        self.cinit_body = torch.empty((lay.NB, lay.idx_cinit.size), dtype=torch.float32, device=device)
        self.blob_tail = torch.empty(lay.idx_tail.size, dtype=dt, device=device)
        self.packed_key = None                      # (flat.data_ptr(), flat._version) the packed blobs were made from
        self.wgs_body = model.wgs_body
        self.wgs_tail = int(os.environ.get("SR_WGS_TAIL", 256))
        self.wgs_head = int(os.environ.get("SR_WGS_HEAD", 256))
        self.part_a = f32(lay.NB * self.wgs_body * lay.slab_a)
        self.part_b = f32(lay.NB * self.wgs_body * lay.slab_b)
        self.part_tail = f32(self.wgs_tail * lay.slab_tail)
        self.part_head = f32(self.wgs_head * lay.slab_head)
        self.loss_part = torch.zeros(self.wgs_tail, dtype=torch.float32, device=device)   # per-workgroup loss sums (loss fold)
        # static part of the C struct
        n = L.WdsrNet()
        n.F, n.NB, n.R, n.dtype, n.mean = lay.F, lay.NB, lay.R, L.DTYPE_CODE[dt], model.image_mean
        n.chan_tab, n.n_chan = self.chan_tab.data_ptr(), lay.chan_tab.shape[0]
        n.bias_tab, n.bias_const, n.n_bias = self.bias_tab.data_ptr(), self.bias_const.data_ptr(), lay.bias_tab.shape[0]
        n.src, n.dsrc = self.src.data_ptr(), self.dsrc.data_ptr()
        n.src_head_off, n.src_body_off = lay.src_head_off, lay.src_body_off
        n.src_body_stride, n.src_tail_off = lay.src_body_stride, lay.src_tail_off
        for k in ("head", "body", "cinit", "tail"):
            setattr(n, "idx_" + k, self.idx[k].data_ptr())
            setattr(n, "n_idx_" + k, self.idx[k].numel())
        n.blob_head, n.blob_body = self.blob_head.data_ptr(), self.blob_body.data_ptr()
        n.cinit_body, n.blob_tail = self.cinit_body.data_ptr(), self.blob_tail.data_ptr()
        n.part_a, n.part_b = self.part_a.data_ptr(), self.part_b.data_ptr()
        n.part_tail, n.part_head = self.part_tail.data_ptr(), self.part_head.data_ptr()
        n.wgs_body, n.wgs_tail, n.wgs_head = self.wgs_body, self.wgs_tail, self.wgs_head
        n.slab_a, n.slab_b, n.slab_tail, n.slab_head = lay.slab_a, lay.slab_b, lay.slab_tail, lay.slab_head
        for k in ("ga", "gb", "gt", "gh"):
            setattr(n, k + "_sidx", self.g[k][0].data_ptr())
            setattr(n, k + "_dst", self.g[k][1].data_ptr())
            setattr(n, "n_" + k, self.g[k][0].numel())
        n.loss_part = self.loss_part.data_ptr()
        # the Adam update may ride on the weight-norm backward if its table rows cover every parameter exactly once
        ct, bt = lay.chan_tab, lay.bias_tab
        cover = np.zeros(int(model.flat.numel()), dtype=np.int32)
        for v_off, g_off, K, _dst in ct:
            cover[v_off:v_off + K] += 1
            cover[g_off] += 1
        for a_, b_, _d in bt:
            cover[a_] += 1
            if b_ >= 0:
                cover[b_] += 1
        n.adam_in_wn_bwd = int(bool((cover == 1).all()))
        if model.nb_split:
            _, n.chan_split, n.bias_split = lay.split_at(model.nb_split)
            n.nb_split = model.nb_split
        self.net = n

    def call_struct(self) -> "L.WdsrNet":
        """a private copy of the C struct for ONE call: forward runs on the caller's thread, backward on autograd's, and
        neither may see the other's per-call fields"""
        return L.WdsrNet.from_buffer_copy(self.net)


class BASIC_MODEL(nn.Module):

    def __init__(self, params):
        super().__init__()
        self.image_mean = float(params.image_mean)
        self.scale = int(params.scale)
        self.remain_blocks = params.num_blocks
        nin = int(params.num_channels)
        f = int(params.num_residual_units)
        nb = int(params.num_blocks)
        self.num_residual_units, self.num_blocks = f, nb
        if nin != 3 or f not in (24, 32) or self.scale not in (2, 3, 4) or nb < 1:
            raise NotImplementedError(
                "MI355X hot path supports num_channels=3, num_residual_units in {24,32}, scale in {2,3,4} "
                f"(got {nin}, {f}, {self.scale}); there is no generic fallback")
        self.hot_dtype = _hot_dtype(params)
        self.wgs_body = int(getattr(params, "hot_wgs_body", os.environ.get("SR_WGS_BODY", 16)))
        self.layout = get_layout(f, nb, self.scale)
        # Data-parallel training: `hot_grad_segments = 2` (or SR_GRAD_SEGMENTS=2) exposes the flat buffer as TWO parameters,
        # `flat_lo` (head, body[0 .. NB/2)) and `flat_hi` (body[NB/2 ..], tail, skip), views of one storage, and runs
        # backward as two autograd nodes: the gradient of `flat_hi` is final after the first one, so DistributedDataParallel
        # (pretrain.py:239) all-reduces it while the early half of the backward still runs.  Default 1: one parameter.
        self.grad_segments = int(getattr(params, "hot_grad_segments", None) or os.environ.get("SR_GRAD_SEGMENTS", 1))
        if self.grad_segments not in (1, 2):
            raise NotImplementedError("hot_grad_segments must be 1 or 2")
        init = self._reference_init()
        # first block of the "late half" of a two-part backward (an even number of blocks: pair launches must not straddle it)
        self.nb_split = nb - 2 * ((nb // 2 + 1) // 2) if nb >= 2 else 0
        if self.grad_segments == 2 and nb >= 2:
            k = self.layout.split_at(self.nb_split)[0]
            self.flat_lo = nn.Parameter(init[:k])
            self.flat_hi = nn.Parameter(init[k:])
            self._flat_master = init                           # plain attribute: the storage both parameters view
        else:
            self.grad_segments = 1
            self.flat = nn.Parameter(init)
        self._dev = {}

    def __getattr__(self, name):
        if name == "flat" and "_flat_master" in self.__dict__:    # two-segment mode: the whole buffer (not a Parameter)
            return self.__dict__["_flat_master"]
        return super().__getattr__(name)

    def _apply(self, fn, recurse=True):
        if self.grad_segments == 2:                          # keep both parameters views of ONE storage across .cuda() / .to()
            with torch.no_grad():
                master = fn(self._flat_master)
            k = self.flat_lo.numel()
            self.flat_lo.data, self.flat_hi.data = master[:k], master[k:]
            for p in (self.flat_lo, self.flat_hi):
                if p.grad is not None:
                    p.grad.data = fn(p.grad.data)
            self._flat_master = master
            return self
        return super()._apply(fn, recurse)

    # ---- the per-device state holds ctypes pointers and device work buffers: never pickled or deep-copied (the reference
    # trainers pickle whole modules, train_video_superresolution.py:306; EMA / best-model copies use deepcopy) ----
    def __getstate__(self):
        d = self.__dict__.copy()
        d["_dev"] = {}
        return d

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_dev" else copy.deepcopy(v, memo)
        if new.grad_segments == 2:                           # deepcopy cloned the three tensors separately: re-tie the views
            k = new.flat_lo.numel()
            new.flat_lo.data, new.flat_hi.data = new._flat_master[:k], new._flat_master[k:]
        return new

    def __setstate__(self, d):
        self.__dict__.update(d)
        if self.grad_segments == 2:
            k = self.flat_lo.numel()
            self._flat_master = torch.cat([self.flat_lo.detach(), self.flat_hi.detach()])
            self.flat_lo.data, self.flat_hi.data = self._flat_master[:k], self._flat_master[k:]

    # ---- initial values exactly as the reference constructs them (same RNG draws, same constants) ----
    def _reference_init(self) -> torch.Tensor:
        lay = self.layout
        flat = torch.zeros(lay.total)
        gains = {"head": 1.0, "tail": 1.0, "skip.0": 1.0}
        for i in range(lay.NB):
            gains[f"body.{i}.body.0"] = 2.0                      # basic_wdsr_b.py:115
            gains[f"body.{i}.body.2"] = 2.0                      # :126
            gains[f"body.{i}.body.3"] = 1 / math.sqrt(lay.NB)    # :136 (res_scale only seeds weight_g)
        for c in lay.convs:
            conv = nn.Conv2d(c.cin, c.cout, c.k)                 # PyTorch-default init of weight_v (and a bias draw)
            off, shape = lay.entries[c.name + ".weight_v"]
            flat[off:off + conv.weight.numel()] = conv.weight.detach().reshape(-1)
            off, _ = lay.entries[c.name + ".weight_g"]
            flat[off:off + c.cout] = gains[c.name]
        return flat                                             # biases stay 0 (:41,63,76,116,127,137)

    # ---- reference-compatible checkpoints ----
    def named_tensors(self):
        """(reference key, view into the flat parameter) for all 3 x (3*num_blocks + 3) tensors"""
        for name, (off, shape) in self.layout.entries.items():
            yield name, self.flat.detach()[off:off + int(np.prod(shape))].view(shape)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        if destination is None:
            destination = OrderedDict()
        for name, view in self.named_tensors():
            destination[prefix + name] = view
        return destination

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        with torch.no_grad():
            for name, view in self.named_tensors():
                key = prefix + name
                if key not in state_dict:
                    missing_keys.append(key)
                    continue
                src = state_dict[key]
                if tuple(src.shape) != tuple(view.shape):
                    error_msgs.append(f"size mismatch for {key}: checkpoint {tuple(src.shape)} vs model {tuple(view.shape)}")
                    continue
                view.copy_(src)
        known = {prefix + n for n in self.layout.entries}
        for key in state_dict:
            if key.startswith(prefix) and key not in known and key not in (prefix + "flat", prefix + "flat_lo", prefix + "flat_hi"):
                unexpected_keys.append(key)

    # ---- execution ----
    def _state(self, device: torch.device) -> _DeviceState:
        key = (device.type, device.index)
        st = self._dev.get(key)
        if st is None or st.blob_head.dtype != self.hot_dtype:
            st = self._dev[key] = _DeviceState(self, device)
        return st

    def _check_input(self, x, *others):
        if not x.is_cuda or not self.flat.is_cuda:
            raise L.HotpathError("BASIC_MODEL (MI355X hot path) needs CUDA/HIP tensors; there is no CPU fallback")
        for t in (self.flat,) + others:
            if t.device != x.device:
                raise L.HotpathError(f"tensors on different devices ({x.device} vs {t.device}): the kernels take raw pointers "
                                     "and would dereference another GPU's memory")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected N x 3 x H x W input, got {tuple(x.shape)}")
        if x.requires_grad:
            raise NotImplementedError("gradient w.r.t. the input image is not on the hot path "
                                      "(the reference trainers never request it)")

    def _forward_impl(self, x, flat, save_acts: bool):
        st = self._state(x.device)
        lay = self.layout
        n, _, h, w = x.shape
        slots = lay.NB + 1 if save_acts else 2
        acts = torch.empty((slots, n, h, w, lay.F), dtype=self.hot_dtype, device=x.device)
        out = torch.empty((n, 3, self.scale * h, self.scale * w), dtype=torch.float32, device=x.device)
        net = st.call_struct()
        net.N, net.H, net.W = n, h, w
        net.flat, net.x, net.acts, net.out = flat.data_ptr(), x.data_ptr(), acts.data_ptr(), out.data_ptr()
        side = None
        if save_acts and self._saves_side_images():
            side = torch.empty(self._side_shape(n, h, w), dtype=self.hot_dtype, device=x.device)
        net.tsave = side.data_ptr() if side is not None else None
        net.dtsave = None
        # opt-in (`model.assume_static_weights = True`): repeated inference with unchanged parameters re-uses the packed
        # weights of the previous call.  Off by default: an in-place write through `.data` does not bump `_version`.
        # Two-segment mode: the optimizer steps `flat_lo` / `flat_hi`, whose version counters are their own (`p.data = view`
        # shares storage, not the counter): the master's `_version` never moves, so the key is built from the segments.
        if self.grad_segments == 2:
            key = (flat.data_ptr(), self.flat_lo._version, self.flat_hi._version)
        else:
            key = (flat.data_ptr(), flat._version)
        static = getattr(self, "assume_static_weights", False) and not save_acts and st.packed_key == key
        flags = (1 if save_acts else 0) | (2 if static else 0)
        with L.device_guard(x.device):
            L.launch("sr_wdsr_net_forward", L.lib().sr_wdsr_net_forward, ctypes.byref(net), flags, L.stream_ptr(x.device))
        st.packed_key = key
        return out, acts, side

    def _saves_side_images(self) -> bool:
        """bf16: the forward / backward-data kernels keep t and dt of every block so the weight-gradient
        kernels need not recompute them (csrc/wdsr_block.h)."""
        lay = self.layout
        return (self.hot_dtype == torch.bfloat16 and lay.NB > 0 and os.environ.get("SR_RECOMPUTE_WGRAD", "0") != "1")

    def _side_shape(self, n, h, w):
        tiles = ((h + 11) // 12) * ((w + 23) // 24)
        return (self.layout.NB, n, tiles, 288, 24 if self.layout.F == 24 else 32)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input(x)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.contiguous().float()
        if self.grad_segments == 2:
            if torch.is_grad_enabled() and (self.flat_lo.requires_grad or self.flat_hi.requires_grad):
                shared = _Shared()
                token = _NetLoFunction.apply(x, self.flat_lo, self, shared)
                return _NetHiFunction.apply(token, self.flat_hi, self, shared)
            return self._forward_impl(x, self.flat.detach(), False)[0]
        if torch.is_grad_enabled() and self.flat.requires_grad:
            return _NetFunction.apply(x, self.flat, self)
        return self._forward_impl(x, self.flat.detach(), False)[0]

    # ---- loss / optimizer epilogue on the hot path; the plain forward() / loss.backward() route above
    # keeps working unchanged ----
    _LOSS_KINDS = {"l1": 1, "charbonnier": 2}

    @staticmethod
    def _gscale(weight: float, numel: int) -> float:
        """upstream gradient / numel as torch's mean-reduced loss backward forms it: an fp32 division of fp32 values"""
        return float(np.float32(weight) / np.float32(numel))

    def _loss_backward_impl(self, x, hr, flat, kind: str, weight: float, net_out=None):
        """forward + backward with the loss folded into the tail backward (no HR gradient tensor, no loss kernels).
        Returns (net struct ready for the Adam call, out, gflat, keepalive)."""
        st = self._state(x.device)
        out, acts, side = self._forward_impl(x, flat, True)
        grads = torch.empty_like(acts)
        gflat = torch.empty_like(flat)
        dtsave = torch.empty_like(side) if side is not None else None
        net = st.call_struct()
        net.N, net.H, net.W = x.shape[0], x.shape[2], x.shape[3]
        net.flat, net.gflat, net.x = flat.data_ptr(), gflat.data_ptr(), x.data_ptr()
        net.acts, net.grads, net.out = acts.data_ptr(), grads.data_ptr(), out.data_ptr()
        net.tsave = side.data_ptr() if side is not None else None
        net.dtsave = dtsave.data_ptr() if dtsave is not None else None
        net.hr, net.loss_kind, net.loss_gscale = hr.data_ptr(), self._LOSS_KINDS[kind], self._gscale(weight, out.numel())
        with L.device_guard(x.device):
            L.launch("sr_wdsr_net_backward", L.lib().sr_wdsr_net_backward, ctypes.byref(net), L.stream_ptr(x.device))
        return net, out, gflat, (acts, grads, side, dtsave)

    def _can_fold(self, node, sr, hr) -> bool:
        """training.L1Loss / L1_Charbonnier_loss on this network's own output: can the loss be folded into the tail backward?"""
        return (self.grad_segments == 1 and sr.is_cuda and hr.is_cuda and hr.device == sr.device and sr.shape == hr.shape
                and sr.dtype == torch.float32 and sr.is_contiguous() and not hr.requires_grad and sr.data_ptr() == node.out_ptr)

    def _backward_folded(self, node, sr, hr, kind: str):
        """the whole backward of the forward recorded in `node` (a _NetFunction context), with d(loss)/d(out) formed inside the
        tail-backward kernel from `out` and `hr` (loss weight 1: the caller's scalar factors arrive through autograd).
        Returns (gflat, keepalive)."""
        x, acts, side = node.x, node.acts, node.tsave
        (flat,) = node.saved_tensors
        st = self._state(x.device)
        grads = torch.empty_like(acts)
        gflat = torch.empty_like(flat)
        dtsave = torch.empty_like(side) if side is not None else None
        net = st.call_struct()
        net.N, net.H, net.W = x.shape[0], x.shape[2], x.shape[3]
        net.flat, net.gflat, net.x = flat.data_ptr(), gflat.data_ptr(), x.data_ptr()
        net.acts, net.grads, net.out = acts.data_ptr(), grads.data_ptr(), sr.data_ptr()
        net.tsave = side.data_ptr() if side is not None else None
        net.dtsave = dtsave.data_ptr() if dtsave is not None else None
        net.hr, net.loss_kind, net.loss_gscale = hr.data_ptr(), self._LOSS_KINDS[kind], self._gscale(1.0, sr.numel())
        with L.device_guard(x.device):
            if st.packed_key != node.packed_key:     # another forward re-packed the per-device blobs meanwhile: pack ours again
                L.launch("sr_wdsr_net_forward", L.lib().sr_wdsr_net_forward, ctypes.byref(net), 4, L.stream_ptr(x.device))
                st.packed_key = node.packed_key
            L.launch("sr_wdsr_net_backward", L.lib().sr_wdsr_net_backward, ctypes.byref(net), L.stream_ptr(x.device))
        return gflat, (grads, dtsave, hr)

    def _check_target(self, x, hr):
        self._check_input(x, hr)
        want = (x.shape[0], 3, self.scale * x.shape[2], self.scale * x.shape[3])
        if tuple(hr.shape) != want:
            raise ValueError(f"target shape {tuple(hr.shape)} != network output shape {want}")
        return x.contiguous().float(), hr.contiguous().float()

    def loss(self, x: torch.Tensor, hr: torch.Tensor, kind: str = "l1", weight: float = 1.0) -> torch.Tensor:
        """weight * L1(model(x), hr) (pretrain.py:73-75) or weight * Charbonnier (train_video_superresolution.py:43-53,
        eps 1e-12) as a differentiable scalar: `model.loss(lr, hr).backward()` fills `flat.grad` exactly like
        `F.l1_loss(model(lr), hr).backward()`, but the loss gradient is formed inside the tail backward kernel."""
        x, hr = self._check_target(x, hr)
        if self.grad_segments == 2:
            raise NotImplementedError("model.loss() is the one-parameter route; with hot_grad_segments = 2 use forward() + a torch "
                                      "loss (DistributedDataParallel hooks need the two gradient nodes)")
        return _NetLossFunction.apply(x, hr, self.flat, self, kind, float(weight))

    DP_OVERLAP_MIN_PARAMS = 4 * 1024 * 1024

    def receptive_halo(self) -> int:
        """LR pixels of context an output pixel depends on, per side: head 3x3 + one 3x3 per block + tail 3x3 (the 5x5 skip
        needs 2) -- what a tile of inference.tiled_forward must carry around its core"""
        return self.num_blocks + 2

    def ddp_bucket_cap_mb(self) -> float:
        """`bucket_cap_mb` for DistributedDataParallel such that each gradient segment is its own bucket (two-segment
        mode): DDP closes a bucket once it has reached the cap, so the cap is half the smaller segment."""
        if self.grad_segments != 2:
            return 25.0
        return 0.5 * min(self.flat_lo.numel(), self.flat_hi.numel()) * 4 / (1 << 20)

    def make_train_state(self, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        """Adam state for train_step (pretrain.py:137 hyper-parameters by default); `state.lr` may be changed between
        steps (MultiStepLR, pretrain.py:139-142)."""
        return AdamState(self.flat, lr, betas, eps)

    def train_step(self, x: torch.Tensor, hr: torch.Tensor, state: "AdamState", kind: str = "l1", weight: float = 1.0,
                   process_group=None, overlap=None):
        """One whole training step -- forward, loss, backward, Adam -- in ONE call into libsr_hotpath.so; `flat` is updated
        in place.  Returns the loss as a device scalar (no host sync; `.item()` it when pretrain.py:82 would).

        Data parallel (`process_group`, or the default group when torch.distributed is initialised with more than one
        rank): what DistributedDataParallel does for pretrain.py:239, as three calls -- forward + the LATE half of the
        backward, an asynchronous RCCL all-reduce (average) of that half's gradient, the EARLY half of the backward
        running underneath it, the second all-reduce, then Adam on the averaged gradient.  Replicas must start equal
        (broadcast `flat` once, as DDP's constructor does)."""
        x, hr = self._check_target(x, hr)
        if state.exp_avg.device != x.device:
            raise L.HotpathError("optimizer state and input on different devices")
        st = self._state(x.device)
        flat = self.flat.detach()
        lay = self.layout
        n, _, h, w = x.shape
        if self.grad_segments == 2:
            raise NotImplementedError("train_step is the single-GPU fused step; with hot_grad_segments = 2 train through "
                                      "DistributedDataParallel (forward / loss / backward / optimizer)")
        acts = torch.empty((lay.NB + 1, n, h, w, lay.F), dtype=self.hot_dtype, device=x.device)
        grads = torch.empty_like(acts)
        out = torch.empty((n, 3, self.scale * h, self.scale * w), dtype=torch.float32, device=x.device)
        gflat = torch.empty_like(flat)
        side = dtsave = None
        if self._saves_side_images():
            side = torch.empty(self._side_shape(n, h, w), dtype=self.hot_dtype, device=x.device)
            dtsave = torch.empty_like(side)
        net = st.call_struct()
        net.N, net.H, net.W = n, h, w
        net.flat, net.gflat, net.x = flat.data_ptr(), gflat.data_ptr(), x.data_ptr()
        net.acts, net.grads, net.out = acts.data_ptr(), grads.data_ptr(), out.data_ptr()
        net.tsave = side.data_ptr() if side is not None else None
        net.dtsave = dtsave.data_ptr() if dtsave is not None else None
        net.hr, net.loss_kind, net.loss_gscale = hr.data_ptr(), self._LOSS_KINDS[kind], self._gscale(weight, out.numel())
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        scal = state.next_scalars()
        import torch.distributed as dist
        pg = process_group                                   # (given explicitly: the data-parallel route even on one rank)
        if pg is None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            pg = dist.group.WORLD
        with L.device_guard(x.device):
            sp = L.stream_ptr(x.device)
            # Data-parallel: `overlap` = all-reduce the late half's gradient under the early half's backward.  That costs two
            # half-depth weight-gradient launches per kernel and a second slab reduction (+29 us at C2) and leaves the second
            # collective exposed anyway; an all-reduce of C2's 0.77 MB over xGMI is latency-bound (tens of microseconds whatever
            # its size), so for small models ONE collective after the whole backward is faster.  Default: overlap from 4 M
            # parameters (16 MB of gradient) on.
            if overlap is None:
                overlap = flat.numel() >= self.DP_OVERLAP_MIN_PARAMS
            if pg is None:
                L.launch("sr_wdsr_net_train_step", L.lib().sr_wdsr_net_train_step, ctypes.byref(net), state.exp_avg.data_ptr(),
                         state.exp_avg_sq.data_ptr(), flat.numel(), ctypes.byref(scal), float(weight) / out.numel(),
                         loss.data_ptr(), sp)
            elif not (overlap and self.nb_split):
                lib = L.lib()
                L.launch("sr_wdsr_net_forward", lib.sr_wdsr_net_forward, ctypes.byref(net), 1, sp)
                L.launch("sr_wdsr_net_backward_part", lib.sr_wdsr_net_backward_part, ctypes.byref(net), 0, sp)
                avg = dist.get_backend(pg) == "nccl"
                dist.all_reduce(gflat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=pg)   # on this stream
                if not avg:
                    gflat.div_(dist.get_world_size(pg))
                L.launch("sr_adam_step", lib.sr_adam_step, flat.data_ptr(), gflat.data_ptr(), state.exp_avg.data_ptr(),
                         state.exp_avg_sq.data_ptr(), flat.numel(), ctypes.byref(scal), st.loss_part.data_ptr(), st.wgs_tail,
                         float(weight) / out.numel(), loss.data_ptr(), sp)
            else:
                k = lay.split_at(self.nb_split)[0]
                lib = L.lib()
                L.launch("sr_wdsr_net_forward", lib.sr_wdsr_net_forward, ctypes.byref(net), 1, sp)
                L.launch("sr_wdsr_net_backward_part", lib.sr_wdsr_net_backward_part, ctypes.byref(net), 1, sp)
                avg = dist.get_backend(pg) == "nccl"          # RCCL averages in the collective; gloo (CPU rehearsals) sums
                op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
                h_hi = dist.all_reduce(gflat[k:], op=op, group=pg, async_op=True)     # runs under the early half
                L.launch("sr_wdsr_net_backward_part", lib.sr_wdsr_net_backward_part, ctypes.byref(net), 2, sp)
                # the early half's all-reduce has nothing left to hide under: issued synchronously it is enqueued on THIS
                # stream (no event hand-off to the collective's own stream and back: ~12 us each way on the GPU timeline)
                if os.environ.get("SR_DP_LAST_ASYNC") == "1":
                    h_lo = dist.all_reduce(gflat[:k], op=op, group=pg, async_op=True)
                    h_hi.wait()
                    h_lo.wait()                               # (stream-side waits: the host does not block)
                else:
                    h_hi.wait()
                    dist.all_reduce(gflat[:k], op=op, group=pg)
                if not avg:
                    gflat.div_(dist.get_world_size(pg))
                L.launch("sr_adam_step", lib.sr_adam_step, flat.data_ptr(), gflat.data_ptr(), state.exp_avg.data_ptr(),
                         state.exp_avg_sq.data_ptr(), flat.numel(), ctypes.byref(scal), st.loss_part.data_ptr(), st.wgs_tail,
                         float(weight) / out.numel(), loss.data_ptr(), sp)
        _bump_version(self.flat)
        st.packed_key = None                         # the blobs no longer match `flat`
        return loss


def _bump_version(t: torch.Tensor):
    """`flat` was written through its raw pointer: tell autograd (in-place version counter) without launching anything"""
    try:
        torch.autograd.graph.increment_version(t)
    except AttributeError:                           # older torch
        t.data.add_(0)


class AdamState:
    """exp_avg / exp_avg_sq / step of torch.optim.Adam for the single flat parameter"""

    def __init__(self, flat: torch.Tensor, lr: float, betas, eps: float):
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        self.step = 0
        self.exp_avg = torch.zeros_like(flat, memory_format=torch.preserve_format).detach()
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)

    def next_scalars(self) -> "L.AdamScalars":
        """the scalars torch.optim.Adam hands its kernels at this step: computed in double, rounded to float"""
        self.step += 1
        bc1 = 1.0 - self.beta1 ** self.step
        bc2 = 1.0 - self.beta2 ** self.step
        return L.AdamScalars(1.0 - self.beta1, self.beta2, 1.0 - self.beta2, math.sqrt(bc2), self.eps, -(self.lr / bc1))

    def state_dict(self):
        return {"step": self.step, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr,
                "betas": (self.beta1, self.beta2), "eps": self.eps}

    def load_state_dict(self, d):
        self.step, self.lr, self.eps = int(d["step"]), float(d["lr"]), float(d["eps"])
        self.beta1, self.beta2 = (float(b) for b in d["betas"])
        self.exp_avg.copy_(d["exp_avg"])
        self.exp_avg_sq.copy_(d["exp_avg_sq"])


class _NetFunction(torch.autograd.Function):
    """Whole-network forward/backward in two C calls.  Saves the block inputs (bf16 or fp32 NHWC); the
    E-wide and L-wide intermediates are recomputed in backward, never stored."""

    @staticmethod
    def forward(ctx, x, flat, model):
        out, acts, tsave = model._forward_impl(x, flat, True)
        ctx.model, ctx.x, ctx.acts, ctx.tsave = model, x, acts, tsave
        ctx.save_for_backward(flat)                  # autograd's version check: an optimizer step between forward and
        ctx.packed_key = model._state(x.device).packed_key   # backward raises instead of mixing old activations with new weights
        ctx.out_ptr, ctx.folded = out.data_ptr(), None        # (training.L1Loss & co. fold the loss into the tail backward)
        return out

    @staticmethod
    def backward(ctx, dout):
        model, x, acts = ctx.model, ctx.x, ctx.acts
        (flat,) = ctx.saved_tensors
        folded = None
        if ctx.folded is not None:
            # a criterion of mobilesuperresolution_amd.training has already run this backward with the loss folded in: `dout` is
            # its zero token (or the token plus the gradient of some OTHER use of the output, which then runs the usual way)
            g0, gloss, token_ptr = ctx.folded
            folded = g0 * gloss
            if dout.data_ptr() == token_ptr and all(st_ == 0 for st_ in dout.stride()):
                return None, folded, None
        st = model._state(x.device)
        dout = dout.contiguous().float()
        grads = torch.empty_like(acts)
        gflat = torch.empty_like(flat)
        net = st.call_struct()
        net.N, net.H, net.W = x.shape[0], x.shape[2], x.shape[3]
        net.flat, net.gflat, net.x = flat.data_ptr(), gflat.data_ptr(), x.data_ptr()
        net.acts, net.grads, net.dout = acts.data_ptr(), grads.data_ptr(), dout.data_ptr()
        dtsave = torch.empty_like(ctx.tsave) if ctx.tsave is not None else None
        net.tsave = ctx.tsave.data_ptr() if ctx.tsave is not None else None
        net.dtsave = dtsave.data_ptr() if dtsave is not None else None
        with L.device_guard(x.device):
            if st.packed_key != ctx.packed_key:      # another forward re-packed the per-device blobs meanwhile: pack ours again
                L.launch("sr_wdsr_net_forward", L.lib().sr_wdsr_net_forward, ctypes.byref(net), 4, L.stream_ptr(x.device))
                st.packed_key = ctx.packed_key
            L.launch("sr_wdsr_net_backward", L.lib().sr_wdsr_net_backward, ctypes.byref(net), L.stream_ptr(x.device))
        return None, (gflat if folded is None else gflat + folded), None


class _Shared:
    """what the two autograd nodes of the two-segment mode hand each other (forward results, the gradient buffers)"""
    __slots__ = ("out", "acts", "tsave", "dtsave", "grads", "gflat", "x", "packed_key", "net")


class _NetLoFunction(torch.autograd.Function):
    """early half (head, body[0 .. nb_split)).  Its forward runs the WHOLE network (one C call) and returns a token; its
    backward is part 2 of sr_wdsr_net_backward_part and runs after _NetHiFunction's."""

    @staticmethod
    def forward(ctx, x, flat_lo, model, shared):
        flat = model.flat.detach()
        shared.out, shared.acts, shared.tsave = model._forward_impl(x, flat, True)
        shared.x, shared.packed_key = x, model._state(x.device).packed_key
        ctx.model, ctx.shared = model, shared
        ctx.save_for_backward(flat_lo)
        return x.new_zeros(1)

    @staticmethod
    def backward(ctx, gtoken):
        model, sh = ctx.model, ctx.shared
        (flat_lo,) = ctx.saved_tensors
        x = sh.x
        with L.device_guard(x.device):
            L.launch("sr_wdsr_net_backward_part", L.lib().sr_wdsr_net_backward_part, ctypes.byref(sh.net), 2, L.stream_ptr(x.device))
        k = flat_lo.numel()
        return None, sh.gflat[:k], None, None


class _NetHiFunction(torch.autograd.Function):
    """late half (body[nb_split ..], tail, skip): backward part 1; its parameter gradient is final when it returns"""

    @staticmethod
    def forward(ctx, token, flat_hi, model, shared):
        ctx.model, ctx.shared = model, shared
        ctx.save_for_backward(flat_hi)
        return shared.out

    @staticmethod
    def backward(ctx, dout):
        model, sh = ctx.model, ctx.shared
        (flat_hi,) = ctx.saved_tensors
        x, acts = sh.x, sh.acts
        st = model._state(x.device)
        flat = model.flat.detach()
        dout = dout.contiguous().float()
        sh.grads = torch.empty_like(acts)
        sh.gflat = torch.empty_like(flat)
        sh.dtsave = torch.empty_like(sh.tsave) if sh.tsave is not None else None
        net = st.call_struct()
        net.N, net.H, net.W = x.shape[0], x.shape[2], x.shape[3]
        net.flat, net.gflat, net.x = flat.data_ptr(), sh.gflat.data_ptr(), x.data_ptr()
        net.acts, net.grads, net.dout = acts.data_ptr(), sh.grads.data_ptr(), dout.data_ptr()
        net.tsave = sh.tsave.data_ptr() if sh.tsave is not None else None
        net.dtsave = sh.dtsave.data_ptr() if sh.dtsave is not None else None
        sh.net = net
        ctx.dout = dout                                       # keep the HR gradient alive until the early half has run
        with L.device_guard(x.device):
            if st.packed_key != sh.packed_key:
                L.launch("sr_wdsr_net_forward", L.lib().sr_wdsr_net_forward, ctypes.byref(net), 4, L.stream_ptr(x.device))
                st.packed_key = sh.packed_key
            L.launch("sr_wdsr_net_backward_part", L.lib().sr_wdsr_net_backward_part, ctypes.byref(net), 1, L.stream_ptr(x.device))
        k = flat.numel() - flat_hi.numel()
        return x.new_zeros(1), sh.gflat[k:], None, None


class _NetLossFunction(torch.autograd.Function):
    """loss(model(x), hr) as one node: forward AND backward of the network run inside forward() (the loss gradient is
    formed in the tail backward kernel, which also emits the loss sums); backward() scales the stored parameter
    gradient by the upstream scalar."""

    @staticmethod
    def forward(ctx, x, hr, flat, model, kind, weight):
        net, out, gflat, _keep = model._loss_backward_impl(x, hr, flat, kind, weight)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        st = model._state(x.device)
        with L.device_guard(x.device):
            L.launch("sr_loss_value", L.lib().sr_loss_value, st.loss_part.data_ptr(), st.wgs_tail, weight / out.numel(),
                     loss.data_ptr(), L.stream_ptr(x.device))
        ctx.save_for_backward(gflat)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (gflat,) = ctx.saved_tensors
        return None, None, gflat * gloss, None, None, None
