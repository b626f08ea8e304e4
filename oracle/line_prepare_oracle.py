"""CPU oracle for the input hand-off.  TEST INFRASTRUCTURE ONLY.

Restates what the reference's data pipeline does to one grey scan before the model sees it
(/root/reference/data/dataset.py:104-135):
    npThum        : width' = min(int(w * max_h / h), max_w); PIL.Image.fromarray(img).resize((width', max_h))
    get_images    : skimage.img_as_float32 (uint8 -> value / 255), right-pad with 1.0 up to max_w
The resize is third-party arithmetic that is not under /root/reference: Pillow (reference pin pillow==10.3.0,
environment.yaml:62; this image ships 12.2.0 -- the 8-bit resampling code is the same in both).  `Image.resize` on a
mode-'L' image defaults to BICUBIC and runs Pillow's src/libImaging/Resample.c:
  * precompute_coeffs: per output index, double-precision bicubic (a = -0.5) weights over a support of 2 * max(scale, 1)
    source pixels, normalised to sum 1, then converted to 22-bit fixed point (round half away from zero);
  * a horizontal pass, then a vertical pass over its uint8 result: sum = 2^21 + sum(pixel * coeff); out =
    clip(sum >> 22, 0, 255).  A pass whose size does not change is skipped.
Pinned by tests/test_line_prepare.py against the installed Pillow itself (bit for bit) and by the fixtures
tests/golden/line_prepare.npz generated from it (tools/make_goldens_line.py)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for box = (0, in_size): (xmin[out], count[out], kk[out][ksize] int)"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmins, counts = np.zeros(out_size, np.int64), np.zeros(out_size, np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        xmins[xx], counts[xx] = xmin, xmax
    return xmins, counts, kk


def _pass(img, out_size):
    """one resampling pass along axis 1 of a uint8 [rows, in_size] array"""
    in_size = img.shape[1]
    xmins, counts, kk = precompute_coeffs(in_size, out_size)
    out = np.empty((img.shape[0], out_size), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        n = counts[xx]
        s = (1 << (PRECISION_BITS - 1)) + src[:, xmins[xx]:xmins[xx] + n] @ kk[xx, :n]
        out[:, xx] = np.clip(s >> PRECISION_BITS, 0, 255)
    return out


def pil_resize_bicubic_u8(img, out_w, out_h):
    """PIL.Image.fromarray(img).resize((out_w, out_h)) for a uint8 grey image [h, w]"""
    h, w = img.shape
    if (out_w, out_h) == (w, h):
        return img.copy()
    cur = img
    if out_w != w:
        cur = _pass(cur, out_w)
    if out_h != h:
        cur = _pass(cur.T.copy(), out_h).T.copy()
    return cur


def thumb_width(h, w, max_h, max_w):
    """npThum (dataset.py:104-111): width after the aspect-preserving resize to height max_h, capped at max_w"""
    return min(int(w * max_h / h), max_w)


def prepare_line(img, max_w, max_h=64):
    """uint8 [h, w] scan -> uint8 [max_h, max_w] (255 = the 1.0 pad); the model reads it as value / 255
    (dataset.py:114-135 with nch = 1, then the uint8 hand-off of htrvt_amd)"""
    h, w = img.shape
    ow = thumb_width(h, w, max_h, max_w)
    out = np.full((max_h, max_w), 255, np.uint8)
    out[:, :ow] = pil_resize_bicubic_u8(img, ow, max_h)
    return out
