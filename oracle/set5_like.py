"""Deterministic synthetic HR images with Set5's shapes (Set5 itself is absent offline).  Shared by the
fixture generator (oracle/make_golden.py:g13_set5_shaped) and the tests, which re-create HR from the seed instead of
storing 5 MB of pixels.  Test infrastructure: nothing in the product path imports this."""
import torch

SET5_SHAPES = [(512, 512), (288, 288), (256, 256), (280, 280), (344, 228)]     # baby, bird, butterfly, head, woman (H, W)


def set5_like_hr(i, hw):
    """image #i in [0, 1], (3, H, W): smooth colour fields + sharp edges + fine texture (bicubic LR loses something)"""
    h, w = hw
    g = torch.Generator().manual_seed(1300 + i)
    low = torch.rand(1, 3, max(h // 32, 2), max(w // 32, 2), generator=g)
    mid = torch.rand(1, 3, max(h // 8, 2), max(w // 8, 2), generator=g)
    img = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False)
    img = 0.7 * img + 0.3 * torch.nn.functional.interpolate(mid, size=(h, w), mode="bilinear", align_corners=False)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    edges = ((torch.sin(xx / 9.0 + i) * torch.cos(yy / 13.0) > 0.3).float() * 0.25)[None, None]
    tex = 0.04 * torch.randn(1, 3, h, w, generator=g)
    return (img + edges + tex).clamp(0, 1)[0]
