"""GPU parity of htrvt_gemm (plain / batched / conv-gather, NT / NN / TN) against
CPU float64 restatements of the same contraction.  Integer-valued operands make
the bf16 and f32 MFMA paths exact, so any fragment-layout error shows as an O(1)
difference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _ops():
    import htrvt_amd  # noqa: F401
    from htrvt_amd import ops
    return ops


def _ints(shape, lo=-3, hi=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).double()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 80, 96), (384, 192, 256), (130, 72, 40), (256, 2304, 768),
                                   (1000, 264, 200), (512, 3072, 768)])
def test_nt_plain_exact(dtype, M, N, K):
    ops = _ops()
    A, B = _ints((M, K), seed=1), _ints((N, K), seed=2)
    ref = A @ B.t()
    a, b = A.to(dtype).cuda(), B.to(dtype).cuda()
    c = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm(a, b, c, dtype=dtype, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    assert torch.equal(c.double().cpu(), ref.to(dtype).double())


@pytest.mark.parametrize("dtype", DT)
def test_nt_epilogue(dtype):
    ops = _ops()
    M, N, K = 192, 136, 128
    g = torch.Generator().manual_seed(3)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    a, b = A.to(dtype), B.to(dtype)
    pre = (a.double() @ b.double().t()) * 0.5 + bias.double()
    ref = F.gelu(pre) + res.to(dtype).double()
    c = torch.empty(M, N, dtype=dtype, device="cuda")
    p = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm(a.cuda(), b.cuda(), c, dtype=dtype, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, alpha=0.5, act=1,
             bias=bias.cuda(), preact=p, residual=res.to(dtype).cuda())
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert (p.double().cpu() - pre).abs().max() < tol
    assert (c.double().cpu() - ref).abs().max() < tol


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N", [(512, 192), (300, 136), (1024, 384)])
def test_nt_colscale_bias_residual_relu_exact(dtype, M, N):
    """eval-mode BatchNorm folded into the launch: C = relu(acc * colscale[n] + bias[n] + residual); integer operands and
    power-of-two scales keep every step exact (values stay below the bf16 integer range)"""
    ops = _ops()
    K = 64
    A, B = _ints((M, K), lo=-2, hi=3, seed=4), _ints((N, K), lo=-2, hi=3, seed=5)
    g = torch.Generator().manual_seed(6)
    scale = torch.tensor([0.25, 0.5, 1.0, -0.5])[torch.randint(0, 4, (N,), generator=g)].double()
    bias = torch.randint(-8, 9, (N,), generator=g).double()
    res = torch.randint(-16, 17, (M, N), generator=g).double()
    a, b = A.to(dtype).cuda(), B.to(dtype).cuda()
    pre = (A @ B.t()) * scale + bias
    for with_res, relu in ((False, False), (False, True), (True, True)):
        ref = pre.to(dtype).double() + (res if with_res else 0.0)      # the staged value is rounded before the residual
        if relu:
            ref = ref.clamp_min(0.0)
        c = torch.empty(M, N, dtype=dtype, device="cuda")
        ops.gemm(a, b, c, dtype=dtype, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, colscale=scale.float().cuda(),
                 bias=bias.float().cuda(), residual=res.to(dtype).cuda() if with_res else None, act=3 if relu else 0)
        assert torch.equal(c.double().cpu(), ref.to(dtype).double()), (with_res, relu)


@pytest.mark.parametrize("dtype", DT)
def test_nn_and_tn_exact(dtype):
    ops = _ops()
    M, N, K = 160, 96, 72
    A, Bm = _ints((M, K), seed=4), _ints((K, N), seed=5)      # NN: B stored [K][N]
    a, b = A.to(dtype).cuda(), Bm.to(dtype).cuda()
    c = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm(a, b, c, dtype=dtype, M=M, N=N, K=K, lda=K, ldb=N, ldc=N, b_layout=ops.MNMAJOR)
    assert torch.equal(c.double().cpu(), (A @ Bm).to(dtype).double())
    # TN with split-K accumulation into float32: C[M][N] = At^T Bt, At [K2][M], Bt [K2][N]
    K2 = 1000
    At, Bt = _ints((K2, M), -2, 3, seed=6), _ints((K2, N), -2, 3, seed=7)
    c32 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    ops.gemm(At.to(dtype).cuda(), Bt.to(dtype).cuda(), c32, dtype=dtype, M=M, N=N, K=K2, lda=M, ldb=N, ldc=N,
             a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR, split_k=4, accumulate=True, c_f32=True)
    assert torch.equal(c32.double().cpu(), At.t() @ Bt)


@pytest.mark.parametrize("dtype", DT)
def test_batched_attention_strides(dtype):
    """S = scale * Q K^T and O = P V on the [B,N,3,h,hd] qkv layout (HTR_VT.py:29-36)."""
    ops = _ops()
    Bn, N, h, hd = 2, 128, 3, 32
    D = h * hd
    qkv = _ints((Bn, N, 3, h, hd), -2, 3, seed=8)
    q, k, v = qkv[:, :, 0].permute(0, 2, 1, 3), qkv[:, :, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3)
    s_ref = q @ k.transpose(-1, -2) * 0.5
    dq = qkv.to(dtype).cuda()
    s = torch.empty(Bn, h, N, N, dtype=torch.float32, device="cuda")
    ops.gemm(dq, dq, s, dtype=dtype, M=N, N=N, K=hd, lda=3 * D, ldb=3 * D, ldc=N, batch=Bn * h, batch_inner=h,
             sA=(N * 3 * D, hd), sB=(N * 3 * D, hd), sC=(h * N * N, N * N), b_off=D, alpha=0.5, c_f32=True)
    assert torch.equal(s.double().cpu(), s_ref)
    P = _ints((Bn, h, N, N), 0, 3, seed=9)
    o_ref = (P @ v).permute(0, 2, 1, 3).reshape(Bn, N, D)
    o = torch.empty(Bn, N, D, dtype=dtype, device="cuda")
    ops.gemm(P.to(dtype).cuda(), dq, o, dtype=dtype, M=N, N=hd, K=N, lda=N, ldb=3 * D, ldc=D, b_layout=ops.MNMAJOR,
             batch=Bn * h, batch_inner=h, sA=(h * N * N, N * N), sB=(N * 3 * D, hd), sC=(N * D, hd), b_off=2 * D)
    assert torch.equal(o.double().cpu(), o_ref.to(dtype).double())


def _pack_fwd(w, cp):      # [Co,Ci,kh,kw] -> [Co][taps][Cpad]
    Co, Ci, kh, kw = w.shape
    out = torch.zeros(Co, kh * kw, cp, dtype=w.dtype)
    out[:, :, :Ci] = w.permute(0, 2, 3, 1).reshape(Co, kh * kw, Ci)
    return out


def _pack_dgrad(w, cp):    # [Co,Ci,kh,kw] -> [Ci][taps][Cpad(Co)]
    Co, Ci, kh, kw = w.shape
    out = torch.zeros(Ci, kh * kw, cp, dtype=w.dtype)
    out[:, :, :Co] = w.permute(1, 2, 3, 0).reshape(Ci, kh * kw, Co)
    return out


CONVS = [  # B, Hi, Wi, Ci, Co, k, stride, pad
    (2, 8, 256, 192, 192, 3, (1, 1), 1),      # layer1-like, exercises the 256x192 LDS-DMA tile
    (2, 8, 128, 192, 384, 3, (2, 2), 1),
    (2, 4, 128, 384, 768, 1, (2, 2), 0),
    (2, 8, 64, 16, 32, 3, (1, 1), 1),
    (2, 16, 64, 64, 64, 3, (2, 1), 1),
    (2, 8, 128, 64, 128, 3, (2, 2), 1),
    (3, 4, 64, 96, 192, 1, (2, 2), 0),
    (1, 2, 256, 192, 192, 3, (1, 1), 1),
    # strided 1x1 downsample convolutions on the streaming kernel (csrc/conv1x1.hip): K = 192 / 384 / padded 96 -> 128, an M tail
    (2, 16, 256, 192, 192, 1, (2, 1), 0),
    (1, 6, 128, 192, 384, 1, (2, 2), 0),
    (2, 4, 256, 96, 192, 1, (2, 2), 0),
    (3, 4, 128, 384, 768, 1, (2, 2), 0),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cfg", CONVS)
def test_conv_fwd_dgrad_wgrad_exact(dtype, cfg):
    ops = _ops()
    Bn, Hi, Wi, Ci, Co, k, stride, pad = cfg
    x = _ints((Bn, Ci, Hi, Wi), -2, 3, seed=10).requires_grad_(True)
    w = _ints((Co, Ci, k, k), -2, 3, seed=11).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=stride, padding=pad)
    dy = _ints(tuple(y.shape), -2, 3, seed=12)
    y.backward(dy)
    geom = ops.ConvGeom(Bn, Hi, Wi, Ci, Co, k, stride, pad)
    M = Bn * geom.Ho * geom.Wo
    cpi, cpo = ops.cpad(Ci, dtype), ops.cpad(Co, dtype)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    wf = _pack_fwd(w.detach(), cpi).to(dtype).cuda()
    yd = torch.empty(Bn, geom.Ho, geom.Wo, Co, dtype=dtype, device="cuda")
    nmt = ops.gemm_num_mtiles(M, Co, dtype, gather=ops.GATHER_CONV_FWD)
    cs = torch.zeros(nmt, 2, Co, dtype=torch.float32, device="cuda")
    ops.gemm(xd, wf, yd, dtype=dtype, M=M, N=Co, K=geom.taps * cpi, lda=Ci, ldb=geom.taps * cpi, ldc=Co,
             gather=ops.GATHER_CONV_FWD, geom=geom, Cpad=cpi, colstats=cs)
    y_nhwc = y.detach().permute(0, 2, 3, 1)
    if dtype == torch.bfloat16 and k == 1 and geom.Wo % (64 if cpi <= 192 else 32) == 0 and not getattr(ops, "_ENV_TILE", 0):
        from htrvt_amd._lib import lib as _l
        assert "conv1x1_fwd_kernel" in _l.htrvt_last_kernel().decode()
    assert torch.equal(yd.double().cpu(), y_nhwc.to(dtype).double())
    assert torch.allclose(cs[:, 0].sum(0).double().cpu(), y_nhwc.reshape(-1, Co).sum(0), rtol=1e-6, atol=1e-3)
    assert torch.allclose(cs[:, 1].sum(0).double().cpu(), (y_nhwc.reshape(-1, Co) ** 2).sum(0), rtol=1e-6, atol=1e-3)
    # dgrad
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    wd = _pack_dgrad(w.detach(), cpo).to(dtype).cuda()
    dxd = torch.empty(Bn, Hi, Wi, Ci, dtype=dtype, device="cuda")
    ops.gemm(dyd, wd, dxd, dtype=dtype, M=Bn * Hi * Wi, N=Ci, K=geom.taps * cpo, lda=Co, ldb=geom.taps * cpo, ldc=Ci,
             gather=ops.GATHER_CONV_DGRAD, geom=geom, Cpad=cpo)
    assert torch.equal(dxd.double().cpu(), x.grad.permute(0, 2, 3, 1).to(dtype).double())
    # strided dgrad again, one launch per input-pixel parity class (bf16 LDS-DMA kernel only)
    sh, sw = stride
    if dtype == torch.bfloat16 and stride != (1, 1) and Bn * Hi * Wi // (sh * sw) > 128:
        res = _ints((Bn, Hi, Wi, Ci), -2, 3, seed=13)
        dx2 = torch.full((Bn, Hi, Wi, Ci), 7.0, dtype=dtype, device="cuda")
        for a in range(sh):
            for b in range(sw):
                nt = sum(1 for dy_ in range(k) if (a + pad - dy_) % sh == 0) * sum(1 for dx_ in range(k) if (b + pad - dx_) % sw == 0)
                Hq, Wq = (Hi - a + sh - 1) // sh, (Wi - b + sw - 1) // sw
                ops.gemm(dyd, wd, dx2, dtype=dtype, M=Bn * Hq * Wq, N=Ci, K=nt * cpo, lda=Co, ldb=geom.taps * cpo, ldc=Ci,
                         gather=ops.GATHER_CONV_DGRAD, geom=geom, Cpad=cpo, cls=(a, b), residual=res.to(dtype).cuda())
        from htrvt_amd._lib import lib
        if "gemm8p" in lib.htrvt_last_kernel().decode():
            # epilogue from the float32 accumulators: product + residual rounded ONCE
            want = (x.grad.permute(0, 2, 3, 1) + res).to(dtype).double()
        else:
            # the LDS-staged bf16 epilogue adds the residual to the bf16-rounded product (documented double rounding)
            want = (x.grad.permute(0, 2, 3, 1).to(dtype).double() + res).to(dtype).double()
        assert torch.equal(dx2.double().cpu(), want)
    # wgrad (split-K, float32 atomics)
    dwp = torch.zeros(geom.taps, cpi, Co, dtype=torch.float32, device="cuda")
    ops.gemm(xd, dyd, dwp, dtype=dtype, M=geom.taps * cpi, N=Co, K=M, lda=Ci, ldb=Co, ldc=Co,
             a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR, gather=ops.GATHER_CONV_WGRAD, geom=geom, Cpad=cpi,
             split_k=3, accumulate=True, c_f32=True)
    assert torch.equal(dwp.double().cpu(), _pack_fwd(w.grad, cpi).permute(1, 2, 0))


# --------------------------------------------------------------------------------------------------------------------
# split-K: float atomics vs the reproducible slab form, explicit XCD-grouped factors (multiples of 8)
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("split_k", [3, 8, 72])
@pytest.mark.parametrize("slabs", [False, True])
def test_conv_wgrad_splitk_exact(dtype, split_k, slabs):
    """layer-1-like conv weight gradient (M = 9*192, N = 192, K = 9216 output pixels): split factors 8 and 72 take the
    XCD-grouped block mapping of the LDS-DMA kernel (gemm_dma_impl.h), `slabs` the ordered two-launch reduction.
    The result is added to a pre-filled C (accumulate)."""
    ops = _ops()
    Bn, Hi, Wi, Ci, Co, k, stride, pad = 2, 8, 576, 192, 192, 3, (1, 1), 1
    x = _ints((Bn, Ci, Hi, Wi), -2, 3, seed=20).requires_grad_(False)
    w = _ints((Co, Ci, k, k), -1, 2, seed=21).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=stride, padding=pad)
    dy = _ints(tuple(y.shape), -2, 3, seed=22)
    y.backward(dy)
    geom = ops.ConvGeom(Bn, Hi, Wi, Ci, Co, k, stride, pad)
    M = Bn * geom.Ho * geom.Wo
    cpi = ops.cpad(Ci, dtype)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    base = _ints((geom.taps, cpi, Co), -5, 6, seed=23)
    dwp = base.float().cuda()
    ws = torch.empty(split_k, geom.taps * cpi, Co, dtype=torch.float32, device="cuda") if slabs else None
    ops.gemm(xd, dyd, dwp, dtype=dtype, M=geom.taps * cpi, N=Co, K=M, lda=Ci, ldb=Co, ldc=Co,
             a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR, gather=ops.GATHER_CONV_WGRAD, geom=geom, Cpad=cpi,
             split_k=split_k, accumulate=True, c_f32=True, splitk_ws=ws)
    assert torch.equal(dwp.double().cpu(), base + _pack_fwd(w.grad, cpi).permute(1, 2, 0))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("split_k", [4, 8, 64])
def test_linear_wgrad_splitk_slabs_exact(dtype, split_k):
    """dW[N][K] += dy^T x (TN, both operands MN-major) with the slab reduction, non-accumulating and accumulating"""
    ops = _ops()
    rows, N, K = 8192, 256, 192
    dy, x = _ints((rows, N), -2, 3, seed=30), _ints((rows, K), -2, 3, seed=31)
    ref = dy.t() @ x
    for acc in (False, True):
        c = torch.full((N, K), 3.0, dtype=torch.float32, device="cuda")
        ws = torch.empty(split_k, N, K, dtype=torch.float32, device="cuda")
        ops.gemm(dy.to(dtype).cuda(), x.to(dtype).cuda(), c, dtype=dtype, M=N, N=K, K=rows, lda=N, ldb=K, ldc=K,
                 a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR, split_k=split_k, accumulate=True, c_f32=True, splitk_ws=ws)
        assert torch.equal(c.double().cpu(), ref + 3.0), acc


# --------------------------------------------------------------------------------------------------------------------
# the conv-dgrad launch the training step spends most of its time in: backward-of-ReLU mask and BatchNorm-backward
# column sums in the staged epilogue (Engine.conv_dgrad: one launch, or one per input-pixel parity class with
# bnb_tile0 offsets for strided convolutions)
# --------------------------------------------------------------------------------------------------------------------
def _sparse_ints(shape, seed, p=0.25):
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(-1, 2, shape, generator=g) * (torch.rand(shape, generator=g) < p)).double()


LAST_FUSED_KERNEL = ""
FUSED_DGRAD = [  # B, Hi, Wi, Ci, Co, k, stride, pad
    (2, 8, 256, 192, 192, 3, (1, 1), 1),     # layer-1 body: 16 full M tiles
    (1, 5, 100, 192, 192, 3, (1, 1), 1),     # M = 500: tail tile
    (2, 16, 128, 192, 192, 3, (2, 1), 1),    # layer1.0.conv1: 2 parity classes
    (2, 8, 144, 192, 384, 3, (2, 2), 1),     # layer2.0.conv1: 4 parity classes, M per class = 576 (tail tile)
    (2, 8, 144, 192, 384, 1, (2, 2), 0),     # downsample 1x1: three of the four classes receive no tap
]


@pytest.mark.parametrize("cfg", FUSED_DGRAD)
@pytest.mark.parametrize("nbn", [1, 2])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_dgrad_fused_relu_bn_sums_exact(cfg, nbn, with_res):
    import htrvt_amd
    from htrvt_amd.engine import Engine, ModelShape
    ops = _ops()
    dtype = torch.bfloat16
    Bn, Hi, Wi, Ci, Co, k, stride, pad = cfg
    eng = Engine(ModelShape(80, (64, 512), 64, 2, 2), dtype, "cuda")
    geom = ops.ConvGeom(Bn, Hi, Wi, Ci, Co, k, stride, pad)
    # sparse +-1 operands keep |dx| far below 256, so the bf16 result and every float32 sum are exact integers
    w = _sparse_ints((Co, Ci, k, k), 40)
    dy = _sparse_ints((Bn, Co, geom.Ho, geom.Wo), 41)
    dx = torch.nn.grad.conv2d_input((Bn, Ci, Hi, Wi), w, dy, stride=stride, padding=pad).permute(0, 2, 3, 1)   # NHWC
    res = _ints((Bn, Hi, Wi, Ci), -3, 4, seed=42)
    relu_src = _ints((Bn, Hi, Wi, Ci), -1, 2, seed=43)            # a third of the elements pass
    g_ref = (dx + (res if with_res else 0.0)) * (relu_src > 0)
    assert g_ref.abs().max() < 256
    gen = torch.Generator().manual_seed(44)
    bnx = [_ints((Bn, Hi, Wi, Ci), -4, 5, seed=45 + t) for t in range(nbn)]
    mean = [torch.randint(-2, 3, (Ci,), generator=gen).double() for _ in range(nbn)]
    rstd = [torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (Ci,), generator=gen)].double() for _ in range(nbn)]

    cpo = ops.cpad(Co, dtype)
    wd = _pack_dgrad(w, cpo).to(dtype).cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    rows = eng.dgrad_tiles(geom)
    parts = [torch.full((rows, 2, Ci), float("nan"), dtype=torch.float32, device="cuda") for _ in range(nbn)]
    bnb = [(bnx[t].to(dtype).cuda(), mean[t].float().cuda(), rstd[t].float().cuda(), parts[t]) for t in range(nbn)]
    if nbn == 1 and not with_res and not eng._dgrad_by_class(geom):
        # the same mask recomputed from the BatchNorm input (HtrvtGemmDesc.relu_scale / relu_shift): a1 = relu(x * sc + sf)
        sc = torch.tensor([0.5, 1.0, -1.0, 2.0])[torch.randint(0, 4, (Ci,), generator=gen)].double()
        sf = torch.randint(-3, 4, (Ci,), generator=gen).double()
        g2_ref = dx * ((bnx[0] * sc + sf) > 0)
        parts2 = torch.full((rows, 2, Ci), float("nan"), dtype=torch.float32, device="cuda")
        out2 = eng.conv_dgrad(dyd, wd, geom, bnb=[(bnb[0][0], bnb[0][1], bnb[0][2], parts2)], relu_bn=(sc.float().cuda(), sf.float().cuda()))
        assert torch.equal(out2.double().cpu(), g2_ref), float((out2.double().cpu() - g2_ref).abs().max())
        xhat = (bnx[0] - mean[0]) * rstd[0]
        gg, xx = g2_ref.reshape(-1, Ci), xhat.reshape(-1, Ci)
        want = torch.stack([torch.stack([gg[r0:r0 + 256].sum(0), (gg[r0:r0 + 256] * xx[r0:r0 + 256]).sum(0)]) for r0 in range(0, gg.shape[0], 256)])
        assert torch.equal(parts2.double().cpu(), want)

    out = eng.conv_dgrad(dyd, wd, geom, residual=res.to(dtype).cuda() if with_res else None,
                         relu_src=relu_src.to(dtype).cuda(), bnb=bnb)
    global LAST_FUSED_KERNEL        # which kernel served the launch above (tests/test_gemm8p_gpu.py forces families and checks)
    from htrvt_amd._lib import lib as _l
    LAST_FUSED_KERNEL = _l.htrvt_last_kernel().decode()
    assert torch.equal(out.double().cpu(), g_ref), float((out.double().cpu() - g_ref).abs().max())

    # expected partial rows: 256-row M tiles of every launch, in launch order
    sh, sw = stride
    by_class = eng._dgrad_by_class(geom)
    for t in range(nbn):
        xhat = (bnx[t] - mean[t]) * rstd[t]
        want = []
        classes = [(a, b) for a in range(sh) for b in range(sw)] if by_class else [None]
        for cl in classes:
            gg = g_ref if cl is None else g_ref[:, cl[0]::sh, cl[1]::sw, :]
            xx = xhat if cl is None else xhat[:, cl[0]::sh, cl[1]::sw, :]
            gg, xx = gg.reshape(-1, Ci), xx.reshape(-1, Ci)
            for r0 in range(0, gg.shape[0], 256):
                want.append(torch.stack([gg[r0:r0 + 256].sum(0), (gg[r0:r0 + 256] * xx[r0:r0 + 256]).sum(0)]))
        want = torch.stack(want)
        got = parts[t].double().cpu()
        assert got.shape == want.shape, (got.shape, want.shape)
        assert torch.equal(got, want), (t, float((got - want).abs().max()))

    if nbn == 1 or with_res:
        # the same launches with the ReLU decision as ONE BIT per element (HtrvtGemmDesc.relu_bits; htrvt_bn_apply_mask writes
        # it in the forward pass): bit (i & 7) of byte (i >> 3) over the flattened NHWC tensor -- identical output and sums
        import numpy as np
        mask = torch.from_numpy(np.packbits((relu_src > 0).numpy().reshape(-1), bitorder="little")).cuda()
        parts_b = [torch.full((rows, 2, Ci), float("nan"), dtype=torch.float32, device="cuda") for _ in range(nbn)]
        out_b = eng.conv_dgrad(dyd, wd, geom, residual=res.to(dtype).cuda() if with_res else None, relu_src=mask, relu_bits=True,
                               bnb=[(bnb[t][0], bnb[t][1], bnb[t][2], parts_b[t]) for t in range(nbn)])
        assert torch.equal(out_b, out)
        for t in range(nbn):
            assert torch.equal(parts_b[t], parts[t]), t


@pytest.mark.parametrize("ratio", [1.0, 60.0, 100.0])
def test_conv_dgrad_fused_bn_sums_large_mean(ratio):
    """the fused BatchNorm-backward sums accumulate RAW sum g*x per tile and centre once at the end, (q - mean*a)*rstd, in
    float32: with |channel mean| >> std that is a difference of two large numbers (advisor, round 3).  Random float data
    with channel means of `ratio` standard deviations against float64: the per-tile sums must stay within 1e-3 of
    sum |g * xhat| -- three decimal digits are plenty for a gradient of gamma -- at every ratio."""
    import htrvt_amd
    from htrvt_amd.engine import Engine, ModelShape
    ops = _ops()
    dtype = torch.bfloat16
    Bn, Hi, Wi, Ci, Co = 2, 8, 256, 192, 192
    eng = Engine(ModelShape(80, (64, 512), 64, 2, 2), dtype, "cuda")
    geom = ops.ConvGeom(Bn, Hi, Wi, Ci, Co, 3, (1, 1), 1)
    gen = torch.Generator().manual_seed(7)
    w = _sparse_ints((Co, Ci, 3, 3), 40)
    dy = _sparse_ints((Bn, Co, geom.Ho, geom.Wo), 41)
    dx = torch.nn.grad.conv2d_input((Bn, Ci, Hi, Wi), w, dy, stride=1, padding=1).permute(0, 2, 3, 1)
    relu_src = _ints((Bn, Hi, Wi, Ci), -1, 2, seed=43)
    g_ref = dx * (relu_src > 0)
    std = torch.rand(Ci, generator=gen).double() * 2 + 0.5
    mean = (torch.randint(0, 2, (Ci,), generator=gen).double() * 2 - 1) * ratio * std
    x = (torch.randn(Bn, Hi, Wi, Ci, generator=gen).double() * std + mean).to(dtype)       # the values the kernel reads
    rstd = 1.0 / std
    wd = _pack_dgrad(w, ops.cpad(Co, dtype)).to(dtype).cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    rows = eng.dgrad_tiles(geom)
    part = torch.full((rows, 2, Ci), float("nan"), dtype=torch.float32, device="cuda")
    out = eng.conv_dgrad(dyd, wd, geom, relu_src=relu_src.to(dtype).cuda(),
                         bnb=[(x.cuda(), mean.float().cuda(), rstd.float().cuda(), part)])
    assert torch.equal(out.double().cpu(), g_ref)
    xhat = (x.double() - mean.float().double()) * rstd.float().double()
    gg, xx = g_ref.reshape(-1, Ci), xhat.reshape(-1, Ci)
    want = torch.stack([(gg[r0:r0 + 256] * xx[r0:r0 + 256]).sum(0) for r0 in range(0, gg.shape[0], 256)])
    scale = torch.stack([(gg[r0:r0 + 256] * xx[r0:r0 + 256]).abs().sum(0) for r0 in range(0, gg.shape[0], 256)]).clamp_min(1.0)
    got = part[:, 1].double().cpu()
    err = ((got - want).abs() / scale).max().item()
    print(f"mean / std = {ratio}: worst |sum g xhat error| / sum |g xhat| = {err:.2e}")
    assert err < 1e-3, err


# --------------------------------------------------------------------------------------------------------------------
# strided 3x3 conv dgrad, every parity class in ONE launch on halo-staged tiles (HtrvtGemmDesc.cls_h = -2,
# csrc/gemm_halo_impl.h gemm_halo_s2_kernel): plain, with the 1x1 downsample gradient as A2, and with the fused epilogue
# --------------------------------------------------------------------------------------------------------------------
MERGED_DGRAD = [  # B, Hi, Wi, Ci, Co, stride
    (2, 16, 256, 192, 192, (2, 1)),     # layer1.0.conv1: 2 classes, three column taps per kernel row
    (2, 8, 512, 192, 384, (2, 2)),      # layer2.0.conv1: 4 classes, groups of 1 and 2 k-tiles
    (1, 4, 512, 384, 768, (2, 2)),      # layer3.0.conv1: two N tiles, dY rows hq + 1 past the image
    (1, 2, 512, 64, 128, (2, 2)),       # one coarse row: every dh = +1 row is outside; 128-column tile
]


@pytest.mark.parametrize("cfg", MERGED_DGRAD)
@pytest.mark.parametrize("with_a2", [False, True])
@pytest.mark.parametrize("fused", [0, 1, 2])      # 0: plain; 1: ReLU bit mask + one sum set; 2: residual + ReLU source + two sum sets
def test_conv_dgrad_merged_classes_exact(cfg, with_a2, fused):
    import numpy as np
    import htrvt_amd  # noqa: F401
    from htrvt_amd.engine import Engine, ModelShape
    from htrvt_amd._lib import lib as _l
    ops = _ops()
    dtype = torch.bfloat16
    Bn, Hi, Wi, Ci, Co, stride = cfg
    eng = Engine(ModelShape(80, (64, 512), 64, 2, 2), dtype, "cuda")
    geom = ops.ConvGeom(Bn, Hi, Wi, Ci, Co, 3, stride, 1)
    assert eng._dgrad_merged(geom) > 0
    w = _sparse_ints((Co, Ci, 3, 3), 60)
    wds = _sparse_ints((Co, Ci, 1, 1), 61)
    dy = _sparse_ints((Bn, Co, geom.Ho, geom.Wo), 62)
    dy2 = _sparse_ints((Bn, Co, geom.Ho, geom.Wo), 63)
    dx = torch.nn.grad.conv2d_input((Bn, Ci, Hi, Wi), w, dy, stride=stride, padding=1)
    if with_a2:
        dx = dx + torch.nn.grad.conv2d_input((Bn, Ci, Hi, Wi), wds, dy2, stride=stride, padding=0)
    dx = dx.permute(0, 2, 3, 1)       # NHWC
    cpo = ops.cpad(Co, dtype)
    pair = torch.empty(2, Bn, geom.Ho, geom.Wo, Co, dtype=dtype, device="cuda")       # A2 lies behind A
    pair[0].copy_(dy.permute(0, 2, 3, 1).to(dtype))
    pair[1].copy_(dy2.permute(0, 2, 3, 1).to(dtype))
    if with_a2:
        wd = torch.zeros(Ci, 10, cpo, dtype=dtype)
        wd[:, :9] = _pack_dgrad(w, cpo).to(dtype)
        wd[:, 9:] = _pack_dgrad(wds, cpo).to(dtype)
        wd = wd.cuda()
    else:
        wd = _pack_dgrad(w, cpo).to(dtype).cuda()
    res = _ints((Bn, Hi, Wi, Ci), -3, 4, seed=64)
    relu_src = _ints((Bn, Hi, Wi, Ci), -1, 2, seed=65)
    kw, g_ref, nbn = {}, dx, 0
    rows = eng.dgrad_tiles(geom)
    if fused:
        nbn = fused
        g_ref = (dx + (res if fused == 2 else 0.0)) * (relu_src > 0)
        gen = torch.Generator().manual_seed(66)
        bnx = [_ints((Bn, Hi, Wi, Ci), -4, 5, seed=67 + t) for t in range(nbn)]
        mean = [torch.randint(-2, 3, (Ci,), generator=gen).double() for _ in range(nbn)]
        rstd = [torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (Ci,), generator=gen)].double() for _ in range(nbn)]
        parts = [torch.full((rows, 2, Ci), float("nan"), dtype=torch.float32, device="cuda") for _ in range(nbn)]
        kw["bnb"] = [(bnx[t].to(dtype).cuda(), mean[t].float().cuda(), rstd[t].float().cuda(), parts[t]) for t in range(nbn)]
        if fused == 1:
            kw.update(relu_src=torch.from_numpy(np.packbits((relu_src > 0).numpy().reshape(-1), bitorder="little")).cuda(), relu_bits=True)
        else:
            kw.update(relu_src=relu_src.to(dtype).cuda(), residual=res.to(dtype).cuda())
    assert g_ref.abs().max() < 256
    out = eng.conv_dgrad(pair[0], wd, geom, extra=pair[1] if with_a2 else None, **kw)
    assert "gemm_halo_s2_kernel" in _l.htrvt_last_kernel().decode()
    assert torch.equal(out.double().cpu(), g_ref), float((out.double().cpu() - g_ref).abs().max())
    if fused:
        # partial rows in grid order: coarse row (b, hq) -> classes, heaviest first ((1,1), (1,0), (0,1), (0,0) / (1), (0)) -> segments
        sh, sw = stride
        classes = [(1 - c // sw, (1 - (c & 1)) if sw == 2 else 0) for c in range(2 * sw)]
        for t in range(nbn):
            xhat = (bnx[t] - mean[t]) * rstd[t]
            want = []
            for b in range(Bn):
                for hq in range(Hi // 2):
                    for (a, cb) in classes:
                        for seg in range(Wi // sw // 256):
                            sl = slice(sw * 256 * seg + cb, sw * 256 * (seg + 1), sw)
                            gg, xx = g_ref[b, 2 * hq + a, sl, :], xhat[b, 2 * hq + a, sl, :]
                            want.append(torch.stack([gg.sum(0), (gg * xx).sum(0)]))
            want = torch.stack(want)
            got = parts[t].double().cpu()
            assert got.shape == want.shape, (got.shape, want.shape)
            assert torch.equal(got, want), (t, float((got - want).abs().max()))
