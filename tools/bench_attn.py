#!/usr/bin/env python3
"""Time the fused attention kernels (csrc/attention.hip) at the model shapes, A/B in one process.
    python tools/bench_attn.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import htrvt_amd  # noqa: F401
    from htrvt_amd._lib import check, lib
    from htrvt_amd.ops import ptr, stream
    shapes = [(128, 256, 6, 128), (64, 512, 6, 128), (128, 128, 6, 128), (256, 256, 8, 64)]
    for B, N, h, hd in shapes:
        D = h * hd
        qkv = (torch.randn(B * N, 3 * D, device="cuda") * 1.2).bfloat16()
        dout = torch.randn(B * N, D, device="cuda").bfloat16()
        out = torch.empty(B * N, D, device="cuda", dtype=torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        lse = torch.empty(B * h, N, device="cuda")
        delta = torch.empty_like(lse)
        sc = hd ** -0.5
        bias = dbias = None

        def fwd():
            check(lib.htrvt_attn_fwd(ptr(qkv), ptr(bias), ptr(out), ptr(lse), B, N, h, hd, sc, 1, stream()), "f")

        def bwd():
            check(lib.htrvt_attn_bwd(ptr(qkv), ptr(bias), ptr(out), ptr(dout), ptr(lse), ptr(delta), ptr(dqkv), ptr(dbias), B, N, h, hd, sc, 1, stream()), "b")

        res = []
        for fn, flops, byts in ((fwd, 4.0 * B * h * N * N * hd, 2 * B * N * 4 * D), (bwd, 14.0 * B * h * N * N * hd, 2 * B * N * 12 * D)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            res.append(f"{us:8.1f} us {flops / us / 1e6:7.1f} TF/s {byts / us / 1e6:5.2f} TB/s(min bytes)")
        print(f"B={B} N={N} h={h} hd={hd}: fwd {res[0]} | bwd(dq+dkv, 7 products) {res[1]}", flush=True)


if __name__ == "__main__":
    run()
