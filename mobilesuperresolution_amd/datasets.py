"""Device-side training input pipeline.  Reference: datasets/_isr.py:56-121 -- per item the DataLoader
workers open two PNGs, `_sample_patch` (random crop, HR crop at scale times the LR one), `_augment` (row flip, column flip,
axis swap) and torchvision's `to_tensor`; at >10^3 HR-Mpix/s per GPU eight Python workers (datasets/__init__.py:22-26)
cannot keep up.  Here the decoded uint8 images stay resident in HBM (DIV2K train: 800 HR images, ~7 GB; 288 GB per GPU)
and one kernel (csrc/patches.h) cuts, flips, transposes, converts and normalises a whole batch.  The random draws are
made on the host with the reference's own calls in the reference's order, so a seeded `random.Random` yields exactly
the patches the reference's `__getitem__` yields (tests/test_gpu_input.py)."""
from __future__ import annotations

import random as _random

import numpy as np
import torch

from . import _lib as L

__all__ = ["DevicePatchCache"]

_REC = np.dtype([("lr_off", "<i8"), ("hr_off", "<i8"), ("lr_w", "<i4"), ("hr_w", "<i4"), ("x", "<i4"), ("y", "<i4"),
                 ("flags", "<i4"), ("pad", "<i4")])


def _as_u8(img):
    a = img.cpu().numpy() if isinstance(img, torch.Tensor) else np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected H x W x 3 uint8 images (np.asarray(Image.open(...)), _isr.py:82-84), got {a.dtype} {a.shape}")
    return np.ascontiguousarray(a)


class DevicePatchCache:
    """`params` fields as the reference's dataset reads them: lr_patch_size, scale, ignored_boundary_size, num_patches
    (datasets/_isr.py:22-37,66-67).  `__len__` and the `index // num_patches` item mapping are the TRAIN-mode ones."""

    def __init__(self, lr_images, hr_images, lr_patch_size, scale, ignored_boundary_size=0, num_patches=1, device="cuda"):
        if len(lr_images) != len(hr_images) or not len(lr_images):
            raise ValueError("need as many HR as LR images, and at least one")
        self.P, self.scale = int(lr_patch_size), int(scale)
        self.ignored, self.num_patches = int(ignored_boundary_size), int(num_patches)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.HotpathError("DevicePatchCache (MI355X hot path) keeps its cache in HBM; there is no CPU fallback")
        chunks, self.meta, off = [], [], 0
        for lr, hr in zip(lr_images, hr_images):
            lr, hr = _as_u8(lr), _as_u8(hr)
            if hr.shape[0] < lr.shape[0] * self.scale or hr.shape[1] < lr.shape[1] * self.scale:
                raise ValueError(f"HR {hr.shape} smaller than scale x LR {lr.shape}")
            if min(lr.shape[:2]) - self.P + 1 - 2 * self.ignored <= 0:
                raise ValueError(f"LR image {lr.shape} too small for a {self.P} patch with boundary {self.ignored}")
            self.meta.append((off, lr.shape[0], lr.shape[1], off + lr.size, hr.shape[1]))
            chunks += [lr.reshape(-1), hr.reshape(-1)]
            off += lr.size + hr.size
        self.cache = torch.from_numpy(np.concatenate(chunks)).to(self.device)

    def __len__(self):
        return len(self.meta) * self.num_patches

    def draw(self, index, rng=_random):
        """one item's draws, in the reference's call order: randrange (row), randrange (column) -- _isr.py:90-95 -- then three
        `random() < 0.5` -- :113-121"""
        lr_off, h, w, hr_off, hr_w = self.meta[index // self.num_patches]
        x = rng.randrange(self.ignored, h - self.P + 1 - self.ignored)
        y = rng.randrange(self.ignored, w - self.P + 1 - self.ignored)
        flags = (1 if rng.random() < 0.5 else 0) | (2 if rng.random() < 0.5 else 0) | (4 if rng.random() < 0.5 else 0)
        return (lr_off, hr_off, w, hr_w, x, y, flags, 0)

    def batch(self, indices, rng=_random, want_lr=True, want_hr=True):
        """(lr (B,3,P,P), hr (B,3,P s,P s)) float32 in [0,1] on the device, items in the order of `indices`"""
        recs = np.array([self.draw(i, rng) for i in indices], dtype=_REC)
        b = len(recs)
        with torch.cuda.device(self.device):
            dev_recs = torch.from_numpy(recs.view(np.uint8).reshape(-1)).pin_memory().to(self.device, non_blocking=True)
            lr = torch.empty((b, 3, self.P, self.P), dtype=torch.float32, device=self.device) if want_lr else None
            s = self.P * self.scale
            hr = torch.empty((b, 3, s, s), dtype=torch.float32, device=self.device) if want_hr else None
            L.launch("sr_patch_gather", L.lib().sr_patch_gather, self.cache.data_ptr(), dev_recs.data_ptr(),
                     lr.data_ptr() if lr is not None else None, hr.data_ptr() if hr is not None else None, b, self.P, self.scale,
                     L.stream_ptr())
        return lr, hr
