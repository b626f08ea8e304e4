#!/usr/bin/env python3
"""Diagnostic: each GEMM-shaped piece of the split-bf16 engine against the float32 engine on the same float32 inputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htrvt_amd  # noqa: E402
from htrvt_amd.engine import Engine, ModelShape  # noqa: E402
from htrvt_amd.ops import ConvGeom  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)), float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def main():
    torch.manual_seed(0)
    dev = "cuda"
    sh = ModelShape(80, (64, 512), 64, 2, 2)
    e32 = Engine(sh, torch.float32, dev)
    esp = Engine(sh, torch.float32, dev, split_bf16=True)
    for (M, K, N) in ((512, 64, 192), (512, 256, 64), (4096, 768, 2304)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.1
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        dy = torch.randn(M, N, device=dev)
        pre32, presp = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
        y32 = e32.linear_fwd(x, w, b, act=1, preact=pre32, residual=res)
        ws, wts = esp._lin_w("t%d" % M, w)
        ysp = esp.linear_fwd(x, ws, b, act=1, preact=presp, residual=res)
        print(f"linear_fwd {M}x{N}x{K}: out {rel(ysp, y32)} preact {rel(presp, pre32)}")
        d32 = e32.linear_dgrad(dy, w, None, act=2, preact=torch.randn(M, K, device=dev).mul_(0).add_(pre32[:, :K] if K <= N else 0.3))
        pg = (pre32[:, :K] if K <= N else torch.full((M, K), 0.3, device=dev)).contiguous()
        d32 = e32.linear_dgrad(dy, w, None, act=2, preact=pg)
        dsp = esp.linear_dgrad(dy, ws, wts, act=2, preact=pg)
        print(f"linear_dgrad: {rel(dsp, d32)}")
        dw32, dwsp = torch.zeros(N, K, device=dev), torch.zeros(N, K, device=dev)
        db32, dbsp = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        e32._linear_wgrad(dy, x, dw32, db32)
        esp._linear_wgrad(dy, x, dwsp, dbsp)
        print(f"linear_wgrad: dw {rel(dwsp, dw32)} db {rel(dbsp, db32)}")
    for (B, Hh, Ww, Ci, Co, k, st) in ((4, 8, 512, 16, 16, 3, (1, 1)), (4, 16, 512, 16, 16, 3, (2, 1)), (4, 8, 512, 16, 32, 3, (2, 2)),
                                       (4, 8, 512, 16, 32, 1, (2, 2)), (4, 2, 128, 64, 64, 3, (1, 1)), (8, 8, 1024, 192, 192, 3, (1, 1))):
        g = ConvGeom(B, Hh, Ww, Ci, Co, k, st, 1 if k == 3 else 0)
        x = torch.randn(B, Hh, Ww, Ci, device=dev)
        w = torch.randn(Co, Ci, k, k, device=dev) * 0.1
        dy = torch.randn(B, g.Ho, g.Wo, Co, device=dev)
        res = torch.randn(B, Hh, Ww, Ci, device=dev)
        name = f"c{Ci}{Co}{k}{st}"
        f32p = e32._conv_w(name, w)
        spp = esp._conv_w(name, w)
        y32, cs32, r32 = e32.conv_fwd(x, f32p[0], g, True)
        ysp, cssp, rsp = esp.conv_fwd(x, spp[0], g, True)
        s32 = cs32[:r32].sum(0)
        ssp = cssp[:rsp].sum(0)
        print(f"conv {B}x{Hh}x{Ww} {Ci}->{Co} k{k} s{st}: fwd {rel(ysp, y32)} colstats {rel(ssp, s32)} rows {r32}/{rsp}")
        d32 = e32.conv_dgrad(dy, f32p[1], g, residual=res)
        dsp = esp.conv_dgrad(dy, spp[1], g, residual=res)
        print(f"   dgrad {rel(dsp, d32)}")
        dw32, dwsp = torch.zeros_like(w), torch.zeros_like(w)
        e32._conv_wgrad(dy, x, g, dw32)
        esp._conv_wgrad(dy, x, g, dwsp)
        print(f"   wgrad {rel(dwsp, dw32)}")
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
