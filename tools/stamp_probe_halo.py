#!/usr/bin/env python3
"""Experiment: where a tile of the halo convolution kernels spends its time beside the k loop (needs a build of gemm_halo.hip with
-DHTRVT_EXP_STAMP: consumer wave 0 stamps s_memtime at kernel start, first barrier, end of the k loop, end of the staging phase,
the barrier behind it, end of the epilogue, and after draining its own stores; plus HW_ID for the per-CU timeline):
    cp -r htr-vt_amd/csrc/build htr-vt_amd/csrc/build_stamp && rm htr-vt_amd/csrc/build_stamp/gemm_halo.o
    make -C htr-vt_amd/csrc OBJDIR=$PWD/htr-vt_amd/csrc/build_stamp LIB=$PWD/htr-vt_amd/lib/libhtrvt_stamp.so EXTRA=-DHTRVT_EXP_STAMP
    python tools/stamp_probe_halo.py htr-vt_amd/lib/libhtrvt_stamp.so"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htrvt_amd  # noqa: E402,F401
from htrvt_amd import _lib, ops  # noqa: E402

l = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, argt) in _lib.PROTOTYPES.items():
    fn = getattr(l, name)
    fn.restype, fn.argtypes = res, argt
ops.lib = l
dt = torch.bfloat16
dev = "cuda"
rnd = lambda *s: (torch.rand(*s, device=dev) - 0.5).to(dt)  # noqa: E731


def report(tag, ntiles):
    torch.cuda.synchronize()
    buf = np.zeros(16 * 8192, dtype=np.uint64)
    rc = l.htrvt_debug_read_halo(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    assert rc == 0, rc
    n = min(8192, ntiles)
    st = buf.reshape(8192, 16)[:n].astype(np.int64)
    seg = [(0, 1, "start -> first k-tile landed"), (1, 3, "k loop"), (3, 4, "accumulators -> LDS image"), (4, 5, "barrier"),
           (5, 6, "walk (side loads, stores) + sums"), (6, 7, "own stores drained")]
    print(f"{tag}: {n} tiles; s_memtime ticks of 10 ns: median / p10 / p90")
    for a, b, nme in seg:
        col = st[:, b] - st[:, a]
        print(f"   {nme:36s} {np.median(col) / 100:7.2f} us {np.percentile(col, 10) / 100:7.2f} {np.percentile(col, 90) / 100:7.2f}")
    tot = st[:, 7] - st[:, 0]
    print(f"   {'tile total':36s} {np.median(tot) / 100:7.2f} us; kernel span {(st[:, 7].max() - st[:, 0].min()) / 100:.1f} us")
    key = st[:, 8] & 0xFFFFFF00
    last_end, gaps = {}, []
    for i in np.argsort(st[:, 0]):
        k = int(key[i])
        if k in last_end:
            gaps.append(st[i, 0] - last_end[k])
        last_end[k] = st[i, 7]
    if gaps:
        g = np.array(gaps)
        print(f"   gap on one CU (stores drained -> next tile's start): median {np.median(g) / 100:.2f} us p10 {np.percentile(g, 10) / 100:.2f} p90 {np.percentile(g, 90) / 100:.2f}  ({len(last_end)} CU keys)")


for tag, (B, Hi, Wi, C) in (("l1 192", (128, 8, 1024, 192)), ("l2 384", (128, 4, 512, 384)), ("l3 768", (128, 2, 256, 768))):
    g = ops.ConvGeom(B, Hi, Wi, C, C, 3, (1, 1), 1)
    M = B * Hi * Wi
    x, y = rnd(B, Hi, Wi, C), torch.empty(B, Hi, Wi, C, dtype=dt, device=dev)
    wf, wd = rnd(C, g.taps, C), rnd(C, g.taps, C)
    nmt = ops.gemm_num_mtiles(M, C, dt, gather=ops.GATHER_CONV_FWD)
    cs = torch.empty(nmt, 2, C, dtype=torch.float32, device=dev)
    ntiles = (M // 256) * ((C + 191) // 192)
    for _ in range(3):
        ops.gemm(x, wf, y, dtype=dt, M=M, N=C, K=g.taps * C, lda=C, ldb=g.taps * C, ldc=C, gather=ops.GATHER_CONV_FWD, geom=g, Cpad=C, colstats=cs)
    report(tag + " forward + column sums", ntiles)
    dx = torch.empty(B, Hi, Wi, C, dtype=dt, device=dev)
    kw = dict(dtype=dt, M=M, N=C, K=g.taps * C, lda=C, ldb=g.taps * C, ldc=C, gather=ops.GATHER_CONV_DGRAD, geom=g, Cpad=C)
    for _ in range(3):
        ops.gemm(y, wd, dx, **kw)
    report(tag + " dgrad plain", ntiles)
    res, x0, x1 = rnd(B, Hi, Wi, C), rnd(B, Hi, Wi, C), rnd(B, Hi, Wi, C)
    mean, rstd = torch.rand(C, device=dev) - 0.5, torch.rand(C, device=dev) + 0.5
    p0, p1 = (torch.empty(nmt, 2, C, dtype=torch.float32, device=dev) for _ in range(2))
    bits = torch.randint(0, 256, (M * C // 8,), dtype=torch.uint8, device=dev)
    for _ in range(3):
        ops.gemm(y, wd, dx, residual=res, relu_src=bits, relu_bits=True, bnb=[(x0, mean, rstd, p0), (x1, mean, rstd, p1)], **kw)
    report(tag + " dgrad residual + ReLU bits + 2 sum sets", ntiles)
