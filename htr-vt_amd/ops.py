"""Thin tensor-level wrappers over the C ABI (include/htrvt.h).

PyTorch is plumbing here: it owns device memory and the stream; every wrapper
passes raw device pointers + sizes to libhtrvt_hip.so.  Nothing in this module
computes with torch ops."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import BF16, F32, GATHER_CONV_DGRAD, GATHER_CONV_FWD, GATHER_CONV_WGRAD, KMAJOR, MNMAJOR, GemmDesc, check, lib

import os

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
_ENV_TILE = int(os.environ.get("HTRVT_GEMM_TILE", "0"))


def dt(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {dtype}")


def bk_of(dtype: torch.dtype) -> int:
    return 64 if dtype == torch.bfloat16 else 32


def cpad(c: int, dtype: torch.dtype) -> int:
    b = bk_of(dtype)
    return (c + b - 1) // b * b


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "htrvt ops need device tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


class ConvGeom:
    """Geometry of one NHWC convolution (resnet18.py:6-7,48,59-63)."""

    def __init__(self, B, Hi, Wi, Ci, Co, k, stride, pad):
        self.B, self.Hi, self.Wi, self.Ci, self.Co = B, Hi, Wi, Ci, Co
        self.kh = self.kw = k
        self.sh, self.sw = stride
        self.ph = self.pw = pad
        self.Ho = (Hi + 2 * pad - k) // self.sh + 1
        self.Wo = (Wi + 2 * pad - k) // self.sw + 1
        self.taps = k * k

    def fill(self, d: GemmDesc):
        d.nB, d.Hi, d.Wi, d.Ci, d.Ho, d.Wo, d.Co = self.B, self.Hi, self.Wi, self.Ci, self.Ho, self.Wo, self.Co
        d.kh, d.kw, d.sh, d.sw, d.ph, d.pw = self.kh, self.kw, self.sh, self.sw, self.ph, self.pw


def gemm(A, B, Cout, *, dtype, M, N, K, lda, ldb, ldc, a_layout=KMAJOR, b_layout=KMAJOR, gather=0, geom=None,
         Cpad=0, batch=1, batch_inner=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), split_k=1, alpha=1.0, act=0, c_f32=False,
         accumulate=False, bias=None, colscale=None, preact=None, residual=None, colstats=None, tile=0,
         a_off=0, b_off=0, c_off=0, cls=None, relu_src=None, bnb=None, bnb_tile0=0, splitk_ws=None, a2=None, relu_bn=None,
         relu_bits=False):
    """Enqueue one htrvt_gemm.  A/B/Cout are tensors (only their storage pointer
    is used); *_off are element offsets into them."""
    d = GemmDesc()
    d.dtype = dt(dtype)
    d.a_layout, d.b_layout, d.gather = a_layout, b_layout, gather
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.batch, d.batch_inner = batch, batch_inner
    d.sA_o, d.sA_i = sA
    d.sB_o, d.sB_i = sB
    d.sC_o, d.sC_i = sC
    d.split_k = split_k
    d.splitk_ws = ptr(splitk_ws)    # float32 [split_k][M][N]: reproducible split-K (ordered slab sum, no atomics)
    d.cls_h, d.cls_w = (-1, -1) if cls is None else cls
    d.A2 = ptr(a2)          # class-(0,0) dgrad: the block's 1x1 downsample gradient as one more tap (include/htrvt.h)
    if geom is not None:
        geom.fill(d)
        d.Cpad = Cpad
    d.alpha = alpha
    d.act = act
    d.c_f32 = 1 if c_f32 else 0
    d.accumulate = 1 if accumulate else 0
    d.tile = tile or _ENV_TILE     # HTRVT_GEMM_TILE: force a kernel variant (A/B runs, tests of non-default variants)
    d.bias = ptr(bias)
    d.colscale = ptr(colscale)
    d.preact = ptr(preact)
    d.residual = ptr(residual)
    d.colstats = ptr(colstats)
    d.relu_src = ptr(relu_src)
    d.relu_bits = 1 if relu_bits else 0      # relu_src is the bit mask written by htrvt_bn_apply_mask
    if relu_bn is not None:     # (scale, shift) of the BatchNorm in front of the ReLU: mask recomputed from bnb[0]'s x
        d.relu_scale, d.relu_shift = ptr(relu_bn[0]), ptr(relu_bn[1])
    if bnb:     # [(x, mean, rstd, partial), ...] up to two BatchNorm layers fed by this gradient
        for t, (bx, bm, br, bp) in enumerate(bnb):
            d.bnb_x[t], d.bnb_mean[t], d.bnb_rstd[t], d.bnb_partial[t] = ptr(bx), ptr(bm), ptr(br), ptr(bp)
        d.bnb_tile0 = bnb_tile0
    esz = A.element_size()
    d.A = A.data_ptr() + a_off * esz
    d.B = B.data_ptr() + b_off * B.element_size()
    d.C = Cout.data_ptr() + c_off * Cout.element_size()
    if PROFILE is None:
        check(lib.htrvt_gemm(C.byref(d), stream()), "htrvt_gemm")
        return d
    # bench.py roofline leg: HIP events on the launch stream around this one kernel
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.htrvt_gemm(C.byref(d), stream()), "htrvt_gemm")
    e1.record()
    if geom is None:
        flops = 2.0 * M * N * K * max(batch, 1)
    elif cls is not None and cls[0] == -2:   # every parity class in one launch: the convolution's MACs (+ the 1x1 downsample's with A2)
        flops = 2.0 * geom.B * geom.Ho * geom.Wo * geom.Co * (K // Cpad) * geom.Ci
    elif cls is not None:   # one parity class of a strided dgrad: only its useful MACs
        flops = 2.0 * M * N * (K // Cpad) * geom.Co
    else:   # algorithmic FLOPs of the convolution, whichever of fwd/dgrad/wgrad this launch is
        flops = 2.0 * geom.B * geom.Ho * geom.Wo * geom.Co * geom.taps * geom.Ci
    kern = lib.htrvt_last_kernel().decode()     # in the key: a strided and a stride-1 conv can share (M, N, K) but not the kernel
    key = (d.dtype, a_layout, b_layout, gather, M, N, K, max(batch, 1), kern)
    ent = PROFILE.setdefault(key, {"flops": flops, "events": [], "kernel": kern})
    ent["events"].append((e0, e1))
    return d


def colsum(x, rows, cols, ld, out, *, dti, keep=None, keep_mod=1):
    """out[c] += sum_r x[r*ld + c] (reproducible two-stage sum); x / out may be tensors or raw device addresses"""
    nws = lib.htrvt_colsum_workspace_floats(rows, cols)
    dev = out.device if isinstance(out, torch.Tensor) else x.device
    ws = torch.empty(nws, dtype=torch.float32, device=dev) if nws else None
    xp = x if isinstance(x, int) else ptr(x)
    op = out if isinstance(out, int) else ptr(out)
    check(lib.htrvt_colsum(xp, rows, cols, ld, op, ptr(keep), keep_mod, dti, ptr(ws), stream()), "colsum")


PROFILE = None   # dict while bench.py measures per-kernel durations, else None


def gemm_num_mtiles(M, N, dtype, gather=0, tile=0):
    d = GemmDesc()
    d.dtype = dt(dtype)
    d.M, d.N, d.K = M, N, 1
    d.gather, d.tile = gather, tile
    n = lib.htrvt_gemm_num_mtiles(C.byref(d))
    if n < 0:
        raise RuntimeError(lib.htrvt_last_error().decode())
    return n
