#!/usr/bin/env python3
"""Experiment: per-phase cycle stamps of the LDS-DMA GEMM kernel (needs a build with -DHTRVT_EXP_STAMP):
    make -C htr-vt_amd/csrc OBJDIR=/tmp/ab_stamp LIB=$PWD/htr-vt_amd/lib/ab_stamp.so EXTRA=-DHTRVT_EXP_STAMP
    python tools/stamp_probe.py htr-vt_amd/lib/ab_stamp.so M N K"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htrvt_amd  # noqa: E402,F401
from htrvt_amd import _lib, ops  # noqa: E402

path, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
l = ctypes.CDLL(os.path.abspath(path))
for name, (res, argt) in _lib.PROTOTYPES.items():
    fn = getattr(l, name)
    fn.restype, fn.argtypes = res, argt
ops.lib = l
dt = torch.bfloat16
a = (torch.rand(M, K, device="cuda") - 0.5).to(dt)
b = (torch.rand(N, K, device="cuda") - 0.5).to(dt)
c = torch.empty(M, N, dtype=dt, device="cuda")
for _ in range(3):
    ops.gemm(a, b, c, dtype=dt, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
torch.cuda.synchronize()
nblk = min(8192, ((M + 255) // 256) * ((N + 191) // 192))
buf = np.zeros(16 * 8192, dtype=np.uint64)
rc = l.htrvt_debug_read_bn192(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
assert rc == 0, rc
st = buf.reshape(8192, 16)[:nblk].astype(np.int64)
names = ["start->dma issued", "dma issued->first tile landed", "main loop", "phase1 (acc->LDS)", "barrier", "phase2 (LDS->global)",
         "store drain (vmcnt0)"]
d = np.diff(st[:, :8], axis=1)
print(f"{nblk} workgroups, M={M} N={N} K={K}; s_memtime ticks (100 MHz => 10 ns each), median / p10 / p90 over workgroups")
for i, nme in enumerate(names):
    col = d[:, i]
    print(f"  {nme:32s} {np.median(col):9.0f} {np.percentile(col, 10):9.0f} {np.percentile(col, 90):9.0f}")
for a_, b_, nme in ((5, 9, "phase2 entry -> loop"), (9, 10, "item addressing + ds_read issue"), (10, 11, "wait for LDS reads"),
                    (11, 12, "store issue (4 items)"), (12, 6, "second iteration + exit")):
    col = st[:, b_] - st[:, a_]
    print(f"    {nme:30s} {np.median(col):9.0f} {np.percentile(col, 10):9.0f} {np.percentile(col, 90):9.0f}")
tot = st[:, 7] - st[:, 0]
print(f"  {'total per workgroup':32s} {np.median(tot):9.0f} {np.percentile(tot, 10):9.0f} {np.percentile(tot, 90):9.0f}")
print("  kernel span (first start -> last end):", st[:, 7].max() - st[:, 0].min())
# per-CU timeline: HW_ID -> (se, cu, xcc?) ; gaps between consecutive workgroups on one CU
hw = st[:, 8]
key = hw & 0xFFFFFF00  # drop wave/simd bits (low 8): wave_id[3:0], simd_id[5:4], ...
order = np.argsort(st[:, 0])
last_end = {}
gaps = []
for i in order:
    k = int(key[i])
    if k in last_end:
        gaps.append(st[i, 0] - last_end[k])
    last_end[k] = st[i, 7]
if gaps:
    g = np.array(gaps)
    print(f"  gap between workgroups on one CU (end -> next start): median {np.median(g):.0f} p10 {np.percentile(g, 10):.0f} p90 {np.percentile(g, 90):.0f}  ({len(last_end)} distinct CU keys)")
