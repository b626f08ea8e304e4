#!/usr/bin/env python3
"""Micro-benchmark of htrvt_gemm on the shapes of the HTR-VT step (bf16, B=128, 64x1024).
    python tools/bench_gemm.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htrvt_amd  # noqa: E402
from htrvt_amd import ops  # noqa: E402


LIBS = None
ROUNDS = 1


TILES = []
TILE = 0


def timeit(fn, iters):
    global TILE
    if TILES:
        best = {}
        for _ in range(ROUNDS):
            for t in TILES:
                TILE = t
                best[f"tile{t}"] = min(best.get(f"tile{t}", 1e9), _timeit(fn, iters))
        TILE = 0
        return best
    if LIBS is not None and len(LIBS) > 1:      # interleaved A/B rounds, report min per lib
        best = {}
        for _ in range(ROUNDS):
            for name, l in LIBS.items():
                ops.lib = l
                t = _timeit(fn, iters)
                best[name] = min(best.get(name, 1e9), t)
        return best
    return _timeit(fn, iters)


def _timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--libs", nargs="*", default=[], help="A/B: alternative builds of libhtrvt_hip.so, timed interleaved in this process")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--tiles", nargs="*", type=int, default=[], help="A/B over the `tile` selector of htrvt_gemm (2: DMA BM=128, 3: DMA BM=256)")
    args = ap.parse_args()
    libs = {"default": ops.lib}
    for path in args.libs:
        import ctypes
        from htrvt_amd import _lib
        l = ctypes.CDLL(os.path.abspath(path))
        for name, (res, argt) in _lib.PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, argt
        libs[os.path.basename(path)] = l
    if args.libs:
        libs.pop("default")
    global LIBS, ROUNDS, TILES
    LIBS, ROUNDS, TILES = libs, args.rounds, args.tiles
    _gemm = ops.gemm

    def gemm_t(*a, **kw):
        kw.setdefault("tile", TILE)
        return _gemm(*a, **kw)
    ops.gemm = gemm_t
    dt = torch.bfloat16
    dev = "cuda"
    rnd = lambda *s: (torch.rand(*s, device=dev) - 0.5).to(dt)  # noqa: E731
    rows = []

    def plain(tag, M, N, K, **kw):
        a, b = rnd(M, K), rnd(N, K)
        c = torch.empty(M, N, dtype=dt, device=dev)
        ms = timeit(lambda: ops.gemm(a, b, c, dtype=dt, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, **kw), args.iters)
        rows.append((tag, ms, 2.0 * M * N * K))

    def nn(tag, M, N, K):
        a, b = rnd(M, K), rnd(K, N)
        c = torch.empty(M, N, dtype=dt, device=dev)
        ms = timeit(lambda: ops.gemm(a, b, c, dtype=dt, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, b_layout=ops.MNMAJOR), args.iters)
        rows.append((tag, ms, 2.0 * M * N * K))

    def tn(tag, M, N, K, split):
        a, b = rnd(K, M), rnd(K, N)
        c = torch.zeros(M, N, dtype=torch.float32, device=dev)
        ws = torch.empty(split, M, N, dtype=torch.float32, device=dev) if split > 1 else None    # slab split-K (the engine's default): incl. the ordered sum
        ms = timeit(lambda: ops.gemm(a, b, c, dtype=dt, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, a_layout=ops.MNMAJOR,
                                     b_layout=ops.MNMAJOR, split_k=split, accumulate=True, c_f32=True, splitk_ws=ws), args.iters)
        rows.append((tag, ms, 2.0 * M * N * K))

    def conv(tag, B, Hi, Wi, Ci, Co, k, stride, pad):
        g = ops.ConvGeom(B, Hi, Wi, Ci, Co, k, stride, pad)
        M = B * g.Ho * g.Wo
        x, y = rnd(B, Hi, Wi, Ci), torch.empty(B, g.Ho, g.Wo, Co, dtype=dt, device=dev)
        wf, wd = rnd(Co, g.taps, Ci), rnd(Ci, g.taps, Co)
        nmt = ops.gemm_num_mtiles(M, Co, dt, gather=ops.GATHER_CONV_FWD)
        cs = torch.empty(nmt, 2, Co, dtype=torch.float32, device=dev)
        fl = 2.0 * M * Co * g.taps * Ci
        ms = timeit(lambda: ops.gemm(x, wf, y, dtype=dt, M=M, N=Co, K=g.taps * Ci, lda=Ci, ldb=g.taps * Ci, ldc=Co,
                                     gather=ops.GATHER_CONV_FWD, geom=g, Cpad=Ci, colstats=cs), args.iters)
        rows.append((tag + " fwd", ms, fl))
        dx = torch.empty(B, Hi, Wi, Ci, dtype=dt, device=dev)
        ms = timeit(lambda: ops.gemm(y, wd, dx, dtype=dt, M=B * Hi * Wi, N=Ci, K=g.taps * Co, lda=Co, ldb=g.taps * Co, ldc=Ci,
                                     gather=ops.GATHER_CONV_DGRAD, geom=g, Cpad=Co), args.iters)
        rows.append((tag + " dgrad", ms, fl))
        dw = torch.zeros(g.taps, Ci, Co, dtype=torch.float32, device=dev)
        from htrvt_amd.engine import Engine, ModelShape
        eng_ = Engine(ModelShape(80, (64, 1024), 768, 4, 6), dt)
        split_h = eng_._split_k(g.taps * Ci, Co, M, conv=True, tiling=eng_._hwgrad_tiles(g))    # halo-staged kernel's tiling
        split = eng_._split_k(g.taps * Ci, Co, M, conv=True)                                     # generic kernel
        split = int(os.environ.get('HTRVT_BENCH_SPLITK', split))
        split_h = int(os.environ.get('HTRVT_BENCH_SPLITH', split_h))
        split17 = eng_._split_k(g.taps * Ci, Co, M, conv=True, tiling=(3 * (Ci // 64) * ((Co + 191) // 192), 192, 192))   # tile 17: unpaired 64-channel tiles
        ms = timeit(lambda: ops.gemm(x, y, dw, dtype=dt, M=g.taps * Ci, N=Co, K=M, lda=Ci, ldb=Co, ldc=Co, a_layout=ops.MNMAJOR,
                                     b_layout=ops.MNMAJOR, gather=ops.GATHER_CONV_WGRAD, geom=g, Cpad=Ci,
                                     split_k=split_h if TILE in (0, 13, 18, 19) else (split17 if TILE == 17 else split), accumulate=True, c_f32=True), args.iters)
        rows.append((tag + f" wgrad(split {split_h}/{split})", ms, fl))

    if not args.only or "plain" in args.only:
        plain("NT 4096^3", 4096, 4096, 4096)
        plain("NT 8192x8192x1024", 8192, 8192, 1024)
        plain("NT qkv 32768x2304x768", 32768, 2304, 768)
        plain("NT fc2 32768x768x3072", 32768, 768, 3072)
        nn("NN dgrad-fc1 32768x768x3072", 32768, 768, 3072)
        nn("NN dgrad-qkv 32768x768x2304", 32768, 768, 2304)
        tn("TN wgrad-fc1 3072x768xK32768", 3072, 768, 32768, 8)
    if "enc" in args.only:      # the encoder's Linear layers at B = 128, N = 256 tokens (M = 32768), forward and dgrad (K-major x K-major)
        M, D, F3, F = 32768, 768, 2304, 3072
        bq, bf, bd = torch.rand(F3, device=dev), torch.rand(F, device=dev), torch.rand(D, device=dev)
        res = rnd(M, D)
        pre = torch.empty(M, F, dtype=dt, device=dev)
        pre_in = rnd(M, F)
        plain("NT 4096^3", 4096, 4096, 4096)
        plain("NT 8192^3", 8192, 8192, 8192)
        plain("qkv fwd +bias        32768x2304x768", M, F3, D, bias=bq)
        plain("proj fwd +bias+res   32768x768x768", M, D, D, bias=bd, residual=res)
        plain("fc1 fwd +bias+gelu+pre 32768x3072x768", M, F, D, bias=bf, act=1, preact=pre)
        plain("fc2 fwd +bias+res    32768x768x3072", M, D, F, bias=bd, residual=res)
        plain("fc2 dgrad *gelu'     32768x3072x768", M, F, D, act=2, preact=pre_in)
        plain("fc1 dgrad            32768x768x3072", M, D, F)
        plain("qkv dgrad            32768x768x2304", M, D, F3)
        plain("proj dgrad           32768x768x768", M, D, D)
    if "encsmall" in args.only:   # the encoder's Linear layers at the per-rank batches of a strong-scaling run: M = 4096 (B = 16), 8192 (B = 32)
        D, F3, F = 768, 2304, 3072
        for M in (4096, 8192):
            bq, bf, bd = torch.rand(F3, device=dev), torch.rand(F, device=dev), torch.rand(D, device=dev)
            res = rnd(M, D)
            pre = torch.empty(M, F, dtype=dt, device=dev)
            pre_in = rnd(M, F)
            plain(f"qkv fwd +bias        {M}x2304x768", M, F3, D, bias=bq)
            plain(f"proj fwd +bias+res   {M}x768x768", M, D, D, bias=bd, residual=res)
            plain(f"fc1 fwd +bias+gelu+pre {M}x3072x768", M, F, D, bias=bf, act=1, preact=pre)
            plain(f"fc2 fwd +bias+res    {M}x768x3072", M, D, F, bias=bd, residual=res)
            plain(f"fc2 dgrad *gelu'     {M}x3072x768", M, F, D, act=2, preact=pre_in)
            plain(f"fc1 dgrad            {M}x768x3072", M, D, F)
            plain(f"qkv dgrad            {M}x768x2304", M, D, F3)
    if "lwgrad" in args.only:   # the encoder's Linear weight gradients dW[out][in] = dy^T x, K = 32768 tokens, over split factors
        for tag, Mo, No in (("fc1", 3072, 768), ("fc2", 768, 3072), ("qkv", 2304, 768), ("proj", 768, 768)):
            for split in ((3, 4, 5, 6, 7, 8, 9, 10, 14) if tag != "proj" else (8, 12, 14, 16, 20, 24, 28)):
                tn(f"TN wgrad-{tag} {Mo}x{No}xK32768 split {split}", Mo, No, 32768, split)
    if not args.only or "mlp" in args.only:
        M, D, F = 32768, 768, 3072
        bias = torch.rand(F, device=dev)
        pre = torch.empty(M, F, dtype=dt, device=dev)
        plain("fc1 fwd plain", M, F, D)
        plain("fc1 fwd +bias", M, F, D, bias=bias)
        plain("fc1 fwd +bias+gelu", M, F, D, bias=bias, act=1)
        plain("fc1 fwd +bias+gelu+preact", M, F, D, bias=bias, act=1, preact=pre)
        a, b = rnd(M, D), rnd(D, F)
        c = torch.empty(M, F, dtype=dt, device=dev)
        pre2 = rnd(M, F)
        ms = timeit(lambda: ops.gemm(a, b, c, dtype=dt, M=M, N=F, K=D, lda=D, ldb=F, ldc=F, b_layout=ops.MNMAJOR), args.iters)
        rows.append(("fc2 dgrad plain (NN)", ms, 2.0 * M * F * D))
        ms = timeit(lambda: ops.gemm(a, b, c, dtype=dt, M=M, N=F, K=D, lda=D, ldb=F, ldc=F, b_layout=ops.MNMAJOR, act=2,
                                     preact=pre2), args.iters)
        rows.append(("fc2 dgrad * gelu'(preact)", ms, 2.0 * M * F * D))
    if "ldpad" in args.only:
        for (M, N, K) in ((32768, 3072, 768), (32768, 768, 3072), (32768, 3072, 3072), (32768, 2304, 768), (32768, 768, 768)):
            for pad in (0, 64, 32, 8):
                a, b = rnd(M, K + pad), rnd(N, K + pad)
                c = torch.empty(M, N + pad, dtype=dt, device=dev)
                ms = timeit(lambda: ops.gemm(a, b, c, dtype=dt, M=M, N=N, K=K, lda=K + pad, ldb=K + pad, ldc=N + pad), args.iters)
                rows.append((f"NT {M}x{N}x{K} ld pad {pad}", ms, 2.0 * M * N * K))
    if "ksweep" in args.only:
        for K in (64, 128, 256, 512, 768, 1536, 3072, 6144):
            plain(f"NT 32768x3072 K={K}", 32768, 3072, K)
        for K in (64, 128, 256, 512, 768, 1536, 3072, 6144):
            plain(f"NT 32768x768 K={K}", 32768, 768, K)
    if not args.only or "conv" in args.only:
        conv("l1 192->192 s1 [128,8,1024]", 128, 8, 1024, 192, 192, 3, (1, 1), 1)
        conv("l2 384->384 s1 [128,4,512]", 128, 4, 512, 384, 384, 3, (1, 1), 1)
        conv("l3 768->768 s1 [128,2,256]", 128, 2, 256, 768, 768, 3, (1, 1), 1)
        conv("l2.0 192->384 s2 [128,8,1024]", 128, 8, 1024, 192, 384, 3, (2, 2), 1)
    if "sconv" in args.only:       # the two stride-(2,2) 3x3 convolutions of the stem (conv1 of layer2.0 / layer3.0)
        conv("l2.0 192->384 s2 [128,8,1024]", 128, 8, 1024, 192, 384, 3, (2, 2), 1)
        conv("l3.0 384->768 s2 [128,4,512]", 128, 4, 512, 384, 768, 3, (2, 2), 1)
    if "c1x1" in args.only:        # the three 1x1 strided downsample convolutions of the stem
        conv("l1.0 ds 192->192 s(2,1) [128,16,1024]", 128, 16, 1024, 192, 192, 1, (2, 1), 0)
        conv("l2.0 ds 192->384 s(2,2) [128,8,1024]", 128, 8, 1024, 192, 384, 1, (2, 2), 0)
        conv("l3.0 ds 384->768 s(2,2) [128,4,512]", 128, 4, 512, 384, 768, 1, (2, 2), 0)
    if "s1conv" in args.only:      # the nine 3x3 stride-1 convolutions of the stem (three per stage), forward + dgrad + wgrad
        conv("l1 192->192 s1 [128,8,1024]", 128, 8, 1024, 192, 192, 3, (1, 1), 1)
        conv("l2 384->384 s1 [128,4,512]", 128, 4, 512, 384, 384, 3, (1, 1), 1)
        conv("l3 768->768 s1 [128,2,256]", 128, 2, 256, 768, 768, 3, (1, 1), 1)
    if "head" in args.only:        # the 80-class head (N = 80 padded to 80: float32 logits), forward / dgrad / weight gradient
        M, D, C = 32768, 768, 80
        a, w = rnd(M, D), rnd(C, D)
        c32 = torch.empty(M, C, dtype=torch.float32, device=dev)
        bias = torch.rand(C, device=dev)
        rows.append(("head fwd 32768x80x768 f32 out", timeit(lambda: ops.gemm(a, w, c32, dtype=dt, M=M, N=C, K=D, lda=D, ldb=D, ldc=C, c_f32=True, bias=bias), args.iters), 2.0 * M * C * D))
        cb = torch.empty(M, C, dtype=dt, device=dev)
        rows.append(("head fwd 32768x80x768 bf16 out", timeit(lambda: ops.gemm(a, w, cb, dtype=dt, M=M, N=C, K=D, lda=D, ldb=D, ldc=C, bias=bias), args.iters), 2.0 * M * C * D))
        plain("NT 32768x128x768", M, 128, D)
        dyh, wt = rnd(M, C), rnd(D, C)
        dx = torch.empty(M, D, dtype=dt, device=dev)
        rows.append(("head dgrad 32768x768x80", timeit(lambda: ops.gemm(dyh, wt, dx, dtype=dt, M=M, N=D, K=C, lda=C, ldb=C, ldc=D), args.iters), 2.0 * M * C * D))
        tn("TN head wgrad 80x768xK32768", C, D, M, 8)
        tn("TN head wgrad^T 768x80xK32768", D, C, M, 8)
        tn("TN head wgrad^T 768x80xK32768 split 28", D, C, M, 28)
    if "sdgrad" in args.only:      # the three strided 3x3 conv dgrads of the stem as engine.backward launches them: per parity class vs merged
        from htrvt_amd.engine import Engine, ModelShape
        eng = Engine(ModelShape(80, (64, 1024), 768, 4, 6), dt)
        for tag, (B, Hi, Wi, Ci, Co, st_), fused in (("l1.0 192->192 s(2,1)", (128, 16, 1024, 192, 192, (2, 1)), False),
                                                      ("l2.0 192->384 s(2,2)", (128, 8, 1024, 192, 384, (2, 2)), True),
                                                      ("l3.0 384->768 s(2,2)", (128, 4, 512, 384, 768, (2, 2)), True)):
            g = ops.ConvGeom(B, Hi, Wi, Ci, Co, 3, st_, 1)
            pair = rnd(2, B, g.Ho, g.Wo, Co)
            wd = rnd(Ci, 10, Co)
            fl = 2.0 * B * g.Ho * g.Wo * Co * 10 * Ci
            x0 = rnd(B, Hi, Wi, Ci)
            mean, rstd = torch.rand(Ci, device=dev) - 0.5, torch.rand(Ci, device=dev) + 0.5
            bits = torch.randint(0, 256, (B * Hi * Wi * Ci // 8,), dtype=torch.uint8, device=dev)
            res = {}
            for rnd_ in range(args.rounds):
                for mode in (False, True):
                    eng.merged_strided_dgrad = mode
                    rows_ = eng.dgrad_tiles(g)
                    p0 = torch.empty(rows_, 2, Ci, dtype=torch.float32, device=dev)
                    kw = dict(relu_src=bits, relu_bits=True, bnb=[(x0, mean, rstd, p0)]) if fused else {}
                    t = _timeit(lambda: eng.conv_dgrad(pair[0], wd, g, extra=pair[1], **kw), args.iters)
                    k = "merged" if mode else "per-class"
                    res[k] = min(res.get(k, 1e9), t)
            rows.append((tag + (" dgrad+A2 fused(bits+1set)" if fused else " dgrad+A2 plain"), res, fl))
    if "fdgrad" in args.only:      # the stride-1 conv dgrads with the fused backward epilogues of the stem (engine.backward): side inputs per element
        for tag, (B, Hi, Wi, C) in (("l1 192", (128, 8, 1024, 192)), ("l2 384", (128, 4, 512, 384)), ("l3 768", (128, 2, 256, 768))):
            g = ops.ConvGeom(B, Hi, Wi, C, C, 3, (1, 1), 1)
            M = B * Hi * Wi
            dy, wd, dx = rnd(B, Hi, Wi, C), rnd(C, g.taps, C), torch.empty(B, Hi, Wi, C, dtype=dt, device=dev)
            x0, x1, res, act = rnd(B, Hi, Wi, C), rnd(B, Hi, Wi, C), rnd(B, Hi, Wi, C), rnd(B, Hi, Wi, C)
            mean, rstd = torch.rand(C, device=dev) - 0.5, torch.rand(C, device=dev) + 0.5
            sc, sf = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
            nmt = ops.gemm_num_mtiles(M, C, dt, gather=ops.GATHER_CONV_DGRAD)
            p0, p1 = (torch.empty(nmt, 2, C, dtype=torch.float32, device=dev) for _ in range(2))
            fl = 2.0 * M * C * g.taps * C
            kw = dict(dtype=dt, M=M, N=C, K=g.taps * C, lda=C, ldb=g.taps * C, ldc=C, gather=ops.GATHER_CONV_DGRAD, geom=g, Cpad=C)
            rows.append((f"{tag} dgrad plain", timeit(lambda: ops.gemm(dy, wd, dx, **kw), args.iters), fl))
            rows.append((f"{tag} dgrad relu-from-bn + 1 sum set", timeit(lambda: ops.gemm(dy, wd, dx, bnb=[(x0, mean, rstd, p0)], relu_bn=(sc, sf), **kw), args.iters), fl))
            rows.append((f"{tag} dgrad res + relu + 1 sum set", timeit(lambda: ops.gemm(dy, wd, dx, residual=res, relu_src=act, bnb=[(x0, mean, rstd, p0)], **kw), args.iters), fl))
            bits = torch.randint(0, 256, (M * C // 8,), dtype=torch.uint8, device=dev)
            rows.append((f"{tag} dgrad res + relu BITS + 1 sum set", timeit(lambda: ops.gemm(dy, wd, dx, residual=res, relu_src=bits, relu_bits=True, bnb=[(x0, mean, rstd, p0)], **kw), args.iters), fl))
            rows.append((f"{tag} dgrad res + relu BITS + 2 sum sets", timeit(lambda: ops.gemm(dy, wd, dx, residual=res, relu_src=bits, relu_bits=True, bnb=[(x0, mean, rstd, p0), (x1, mean, rstd, p1)], **kw), args.iters), fl))
            rows.append((f"{tag} dgrad res + relu + 2 sum sets", timeit(lambda: ops.gemm(dy, wd, dx, residual=res, relu_src=act, bnb=[(x0, mean, rstd, p0), (x1, mean, rstd, p1)], **kw), args.iters), fl))
    for tag, ms, fl in rows:
        if isinstance(ms, dict):
            print(f"{tag:42s} " + "  ".join(f"{k}: {v:7.3f} ms {fl / v / 1e9:7.1f} TF" for k, v in ms.items()))
        else:
            print(f"{tag:42s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
