"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's training-item path, datasets/_isr.py:68-121, for the
parity tests of mobilesuperresolution_amd.datasets.  numpy only; `to_tensor` restates torchvision's for uint8 HWC input
(permute to CHW, float32, divide by 255).  PINNED since round 3: oracle/make_golden.py imports the reference's own
ImageSuperResolutionDataset (skimage / torchvision stubbed: third-party, absent here), runs __getitem__ on PNG files under a
seeded `random` and writes fixture G14; tests/test_oracle_golden.py::test_g14_* holds this file to it item for item.
(`to_tensor` itself is torchvision's: restated in the stub, the one unpinned step -- a division by 255.)"""
import numpy as np


def sample_patch(lr_image, hr_image, lr_patch_size, scale, ignored_boundary_size, rng):
    """_isr.py:87-103 (TRAIN)"""
    x = rng.randrange(ignored_boundary_size, lr_image.shape[0] - lr_patch_size + 1 - ignored_boundary_size)
    y = rng.randrange(ignored_boundary_size, lr_image.shape[1] - lr_patch_size + 1 - ignored_boundary_size)
    lr = lr_image[x:x + lr_patch_size, y:y + lr_patch_size]
    hr = hr_image[x * scale:(x + lr_patch_size) * scale, y * scale:(y + lr_patch_size) * scale]
    return lr, hr


def augment(lr, hr, rng):
    """_isr.py:110-122 (TRAIN)"""
    if rng.random() < 0.5:
        lr, hr = lr[::-1], hr[::-1]
    if rng.random() < 0.5:
        lr, hr = lr[:, ::-1], hr[:, ::-1]
    if rng.random() < 0.5:
        lr, hr = np.swapaxes(lr, 0, 1), np.swapaxes(hr, 0, 1)
    return lr, hr


def to_tensor(img):
    """torchvision.transforms.functional.to_tensor on an HWC uint8 array: CHW float32 / 255"""
    return np.ascontiguousarray(img).transpose(2, 0, 1).astype(np.float32) / np.float32(255)


def train_item(lr_images, hr_images, index, lr_patch_size, scale, ignored_boundary_size, num_patches, rng):
    """__getitem__ in TRAIN mode, _isr.py:66-78"""
    i = index // num_patches
    lr, hr = sample_patch(lr_images[i], hr_images[i], lr_patch_size, scale, ignored_boundary_size, rng)
    lr, hr = augment(lr, hr, rng)
    return to_tensor(lr), to_tensor(hr)
