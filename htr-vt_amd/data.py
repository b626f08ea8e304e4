"""Input hand-off in front of the path: the per-image preparation of the reference's loader
(/root/reference/data/dataset.py:104-135 -- PIL resize to height 64 keeping the aspect, img_as_float32, right pad with
1.0 to the model width) as two HIP launches over a ragged batch of raw grey uint8 scans.  The result is the uint8 batch
[B,1,H,W] the model consumes directly (its first kernels read value / 255): a quarter of the host->device bytes of the
float batch, no float image anywhere, bit-identical pixels to the Pillow path."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import check, lib
from .ops import ptr, stream


class _LineImage(C.Structure):      # include/htrvt.h: HtrvtLineImage
    _fields_ = [("src_offset", C.c_int64), ("tmp_offset", C.c_int64), ("h", C.c_int32), ("w", C.c_int32)]


def prepare_lines(images, max_w, max_h=64, device="cuda"):
    """images: sequence of uint8 arrays / CPU tensors [h_i, w_i] (grey scans as `np.array(Image.open(f).convert('L'))`
    delivers them, dataset.py:117).  Returns a uint8 device tensor [B, 1, max_h, max_w]."""
    arrs = [np.ascontiguousarray(im.numpy() if isinstance(im, torch.Tensor) else im) for im in images]
    if not arrs:
        raise ValueError("prepare_lines: empty batch")
    smax = lib.htrvt_line_max_scale()
    table = (_LineImage * len(arrs))()
    so = to = 0
    for i, a in enumerate(arrs):
        if a.dtype != np.uint8 or a.ndim != 2 or a.shape[0] < 1 or a.shape[1] < 1:
            raise ValueError(f"prepare_lines: image {i} must be a 2-D uint8 array, got {a.dtype} {a.shape}")
        h, w = a.shape
        ow = min(int(w * max_h / h), max_w)
        if ow < 1 or h > smax * max_h or w > smax * ow:
            raise ValueError(f"prepare_lines: image {i} ({h}x{w}) shrinks by more than {smax}x (or to zero width)")
        table[i] = _LineImage(so, to, h, w)
        so += h * w
        to += h * max_w
    dev = torch.device(device)
    host = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs]))
    src = host.to(dev, non_blocking=False)
    tab = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
    tmp = torch.empty(to, dtype=torch.uint8, device=dev)
    dst = torch.empty(len(arrs), 1, max_h, max_w, dtype=torch.uint8, device=dev)
    check(lib.htrvt_line_prepare(ptr(src), ptr(tab), ptr(tmp), ptr(dst), len(arrs), max_h, max_w, max(a.shape[0] for a in arrs),
                                 stream()), "line_prepare")
    return dst
