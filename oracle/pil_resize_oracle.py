"""CPU restatement of Pillow's 8-bit BICUBIC resize (libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) -- TEST INFRASTRUCTURE ONLY.

The reference's image transform (inference.py:111-132: Resize(crop, BICUBIC) -> CenterCrop -> ToTensor ->
Normalize) runs on PIL images, so torchvision's Resize IS Pillow's resize.  Pillow (pinned here: 12.2.0, the
version in this image; the algorithm is unchanged since 3.x) is a third-party dependency of the reference;
this file restates its published algorithm in numpy and tests/test_preprocess.py pins it bit-exactly against
Pillow itself on random images.  The HIP kernel (csrc/kernels_preproc.hip) is then checked against both."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = -x if x < 0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support=2.0):
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    # normalize_coeffs_8bpc: round half away from zero into 8.22 fixed point
    ik = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64)
    return ik.astype(np.int32), bounds, ksize


def _pass(img, ik, bounds, axis):
    """img uint8 [H, W, C]; resample along `axis` (1 = horizontal, 0 = vertical)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)          # [in, other, C]
    out = np.empty((ik.shape[0],) + src.shape[1:], dtype=np.uint8)
    for xx in range(ik.shape[0]):
        xmin, xmax = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(ik[xx, :xmax].astype(np.int64), src[xmin:xmin + xmax], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img, out_h, out_w):
    """== np.asarray(Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC)); horizontal pass first."""
    h, w, _ = img.shape
    if w != out_w:
        ik, b, _ = precompute_coeffs(w, out_w)
        img = _pass(img, ik, b, axis=1)
    if h != out_h:
        ik, b, _ = precompute_coeffs(h, out_h)
        img = _pass(img, ik, b, axis=0)
    return img


def reference_transform_size(w, h, crop):
    """torchvision Resize(int): shorter side -> crop, the other int(crop * long / short) (inference.py:119)."""
    if w <= h:
        return crop, int(crop * h / w)
    return int(crop * w / h), crop
