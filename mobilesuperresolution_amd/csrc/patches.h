// Training patches cut on device from a resident uint8 image cache (gfx950).  Reference ops replaced, per patch:
// ImageSuperResolutionDataset._sample_patch (crop at row x, column y; the HR crop at scale times that),
// _augment (flip rows, flip columns, swap the two axes -- in that order) and torchvision's to_tensor (HWC uint8 ->
// CHW float32 / 255), datasets/_isr.py:68-121.  The draws (image, x, y, three coin flips) stay on the host, in the
// reference's own RNG call order; one 40-byte record per patch tells the kernel what to cut.
#pragma once
#include "sr_common.h"

struct PatchRec { long lr_off, hr_off; int lr_w, hr_w, x, y, flags, pad_; };   // flags: 1 flip rows, 2 flip columns, 4 transpose

// out[b][c][i][j] = img[(x + r) * w + (y + q)][c] / 255 with (r, q) = flips(transpose ? (j, i) : (i, j))
__global__ __launch_bounds__(256) void sr_patch_gather_kernel(const unsigned char* __restrict__ cache, const PatchRec* __restrict__ recs,
                                                              float* __restrict__ out, int P, int scale, int hr) {
  const PatchRec rc = recs[blockIdx.y];
  const int S = hr ? P * scale : P;
  const long off = hr ? rc.hr_off : rc.lr_off;
  const int w = hr ? rc.hr_w : rc.lr_w, x0 = hr ? rc.x * scale : rc.x, y0 = hr ? rc.y * scale : rc.y;
  const int total = 3 * S * S;
  float* o = out + (size_t)blockIdx.y * total;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int c = e / (S * S), ij = e - c * S * S, i = ij / S, j = ij - i * S;
    int r = (rc.flags & 4) ? j : i, q = (rc.flags & 4) ? i : j;
    if (rc.flags & 1) r = S - 1 - r;
    if (rc.flags & 2) q = S - 1 - q;
    o[e] = (float)cache[off + ((size_t)(x0 + r) * w + (y0 + q)) * 3 + c] / 255.0f;
  }
}
