"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the image preprocessing in front of the swap path.

Reference call sites: data/base_dataset.py:85-171 (get_transform: __scale_shortside -> PIL
``img.resize(..., Image.BICUBIC)``, __make_power_2 (round to a multiple of 16, resize again),
transforms.ToTensor, transforms.Normalize(0.5, 0.5)).  The arithmetic lives in a third-party
dependency, Pillow (requirements.txt; 12.2.0 in this image): ``ImagingResample`` for 8-bit images
(src/libImaging/Resample.c) -- restated here from its published algorithm: per output sample a
window of ``support * max(scale, 1)`` input samples, bicubic (a = -0.5) weights evaluated in double
at pixel centres, normalised to sum 1, converted to 22-bit fixed point with round-half-away, integer
accumulation from 2^21, arithmetic shift, clamp to [0, 255]; horizontal pass first, then vertical,
each rounding to uint8.  Pinned bit-exactly against Pillow itself by tests/test_image_cpu.py
(Pillow is importable here and on the GPU box).  Plain Python loops: small cases only.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeffs(in_size, out_size, support=2.0, filt=bicubic):
    """precompute_coeffs + normalize_coeffs_8bpc for the full-image box (0, in_size)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - sup + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + sup + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        k += [0.0] * (ksize - xmax)
        bounds.append((xmin, xmax))
        kk.append([int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in k])
    return ksize, bounds, kk


def _clip8(v):
    v >>= PRECISION_BITS       # Python's >> on negative ints is arithmetic, as in C for this use
    return 0 if v < 0 else (255 if v > 255 else v)


def resize_u8(img, out_h, out_w):
    """img (H, W, C) uint8 -> (out_h, out_w, C) uint8, Image.resize((out_w, out_h), Image.BICUBIC)."""
    H, W, C = img.shape
    a = img.astype(np.int64)
    if out_w != W:
        _, bnd, kk = coeffs(W, out_w)
        t = np.zeros((H, out_w, C), dtype=np.int64)
        for xx in range(out_w):
            xmin, n = bnd[xx]
            acc = np.full((H, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for x in range(n):
                acc += a[:, xmin + x, :] * kk[xx][x]
            t[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        a = t
    if out_h != H:
        _, bnd, kk = coeffs(H, out_h)
        t = np.zeros((out_h, a.shape[1], C), dtype=np.int64)
        for yy in range(out_h):
            ymin, n = bnd[yy]
            acc = np.full((a.shape[1], C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for y in range(n):
                acc += a[ymin + y, :, :] * kk[yy][y]
            t[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
        a = t
    return a.astype(np.uint8)


def scale_shortside_size(ow, oh, target_width):
    """__scale_shortside (base_dataset.py:164-168): Python round() = round half to even."""
    scale = target_width / min(ow, oh)
    return round(ow * scale), round(oh * scale)


def make_power_2_size(ow, oh, base=16):
    """__make_power_2 (base_dataset.py:141-149)."""
    return int(round(ow / base) * base), int(round(oh / base) * base)


def preprocess(img, load_size):
    """scale_shortside -> make_power_2 -> ToTensor -> Normalize(0.5, 0.5): (H,W,3) uint8 -> (3,h,w) float32."""
    oh, ow = img.shape[:2]
    w1, h1 = scale_shortside_size(ow, oh, load_size)
    x = resize_u8(img, h1, w1) if (w1, h1) != (ow, oh) else img
    w2, h2 = make_power_2_size(w1, h1)
    if (w2, h2) != (w1, h1):
        x = resize_u8(x, h2, w2)
    t = x.astype(np.float32) / np.float32(255)            # ToTensor: .div(255)
    t = (t - np.float32(0.5)) / np.float32(0.5)           # Normalize
    return np.ascontiguousarray(t.transpose(2, 0, 1))
