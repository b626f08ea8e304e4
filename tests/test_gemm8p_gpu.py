"""The 8-phase bfloat16 GEMM kernels (csrc/gemm8p_impl.h) behind htrvt_gemm: both tile widths forced through the `tile`
selector (10: 256 columns, 2 x 4 waves; 11: 192 columns, 4 x 2 waves) on exact integer data -- tails in M, N and K,
the direct-from-accumulator epilogues, the implicit-GEMM convolution gathers with their fused backward epilogues --
and the older LDS-DMA family (tile 3 / 4) kept green beside it, since it still serves the MN-major and float32-output
shapes.  References: float64 torch on the host (reference call sites: HTR_VT.py:22-37,76 Linear / timm Mlp,
resnet18.py:26-31,59-63 convolutions)."""
import pytest
import torch
import torch.nn.functional as F

import test_gemm_gpu as T

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _last_kernel():
    from htrvt_amd._lib import lib
    return lib.htrvt_last_kernel().decode()


@pytest.fixture
def force_tile(monkeypatch):
    def set_(tile):
        from htrvt_amd import ops
        monkeypatch.setattr(ops, "_ENV_TILE", tile)
    return set_


@pytest.mark.parametrize("tile", [10, 11])
@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (1000, 264, 200), (512, 3072, 768), (300, 96, 72), (4096, 2304, 64),
                                   (260, 24, 1024), (129, 192, 8)])
def test_plain_exact(tile, M, N, K):
    ops = T._ops()
    A, B = T._ints((M, K), seed=1), T._ints((N, K), seed=2)
    ref = A @ B.t()
    a, b = A.to(BF).cuda(), B.to(BF).cuda()
    c = torch.full((M, N), 7.0, dtype=BF, device="cuda")
    ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=tile)
    assert "gemm8p_kernel" in _last_kernel(), _last_kernel()
    assert torch.equal(c.double().cpu(), ref.to(BF).double())


@pytest.mark.parametrize("tile", [10, 11])
def test_plain_padded_leading_dimensions_and_batch(tile):
    """lda / ldb / ldc larger than the row, a batch of independent products (batch strides), nothing outside C's columns written"""
    ops = T._ops()
    M, N, K, nb = 384, 192, 136, 3
    A, B = T._ints((nb, M, K + 8), seed=3), T._ints((nb, N, K + 16), seed=4)
    c = torch.full((nb, M, N + 24), 5.0, dtype=BF, device="cuda")
    ops.gemm(A.to(BF).cuda(), B.to(BF).cuda(), c, dtype=BF, M=M, N=N, K=K, lda=K + 8, ldb=K + 16, ldc=N + 24, batch=nb,
             batch_inner=1, sA=(M * (K + 8), 0), sB=(N * (K + 16), 0), sC=(M * (N + 24), 0), tile=tile)
    assert "gemm8p_kernel" in _last_kernel()
    ref = A[:, :, :K] @ B[:, :, :K].transpose(1, 2)
    got = c.double().cpu()
    assert torch.equal(got[:, :, :N], ref.to(BF).double())
    assert (got[:, :, N:] == 5.0).all()


@pytest.mark.parametrize("tile", [10, 11])
def test_linear_epilogues(tile):
    """bias + exact-erf GELU + saved pre-activation (fc1 forward), bias + residual (proj / fc2 forward), * GELU'(saved)
    (fc2 dgrad): random data against float64"""
    ops = T._ops()
    M, N, K = 520, 384, 192
    g = torch.Generator().manual_seed(3)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    a, b = A.to(BF), B.to(BF)
    acc = a.double() @ b.double().t()
    # GELU + preact
    pre = acc * 0.5 + bias.double()
    c = torch.empty(M, N, dtype=BF, device="cuda")
    p = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm(a.cuda(), b.cuda(), c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, alpha=0.5, act=1, bias=bias.cuda(), preact=p, tile=tile)
    assert "gemm8p_kernel" in _last_kernel()
    pq = p.double().cpu()
    assert (pq - pre).abs().max() <= 2.0 ** -8 * pre.abs().max()                    # the saved value is the rounded pre-activation
    assert (c.double().cpu() - F.gelu(pq)).abs().max() <= 2.0 ** -8 * pq.abs().max() + 2e-6    # GELU of the SAVED value, rounded once
    # GELU without a saved pre-activation
    c2 = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm(a.cuda(), b.cuda(), c2, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, alpha=0.5, act=1, bias=bias.cuda(), tile=tile)
    assert torch.equal(c2, c)
    # residual: ONE rounding of acc + bias + residual
    r = res.to(BF)
    ops.gemm(a.cuda(), b.cuda(), c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias.cuda(), residual=r.cuda(), tile=tile)
    assert "gemm8p_kernel" in _last_kernel()
    want = acc + bias.double() + r.double()
    assert (c.double().cpu() - want).abs().max() <= 2.0 ** -8 * want.abs().max() * 1.01
    # * GELU'(saved pre-activation)
    xs = (torch.randn(M, N, generator=g) * 1.5).to(BF)
    ops.gemm(a.cuda(), b.cuda(), c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, act=2, preact=xs.cuda(), tile=tile)
    assert "gemm8p_kernel" in _last_kernel()
    x64 = xs.double()
    gp = 0.5 * (1 + torch.erf(x64 / 2 ** 0.5)) + x64 * torch.exp(-0.5 * x64 * x64) / (2 * torch.pi) ** 0.5
    want = acc * gp
    assert (c.double().cpu() - want).abs().max() <= 2.0 ** -7 * want.abs().max()


@pytest.mark.parametrize("tile", [10, 11, 4])
@pytest.mark.parametrize("cfg", T.CONVS)
def test_conv_gathers_exact(force_tile, tile, cfg):
    """implicit-GEMM conv forward (+ BatchNorm column sums), dgrad, parity-class dgrad + residual: the existing exact test with
    the kernel family forced (shapes a width cannot serve fall back inside htrvt_gemm and stay exact)"""
    force_tile(tile)
    T.test_conv_fwd_dgrad_wgrad_exact(BF, cfg)


@pytest.mark.parametrize("tile", [0, 11, 4])
@pytest.mark.parametrize("cfg", T.FUSED_DGRAD)
@pytest.mark.parametrize("nbn,with_res", [(1, False), (1, True), (2, True)])
def test_conv_dgrad_fused_epilogues(force_tile, tile, cfg, nbn, with_res):
    force_tile(tile)
    T.test_conv_dgrad_fused_relu_bn_sums_exact(cfg, nbn, with_res)
    if tile == 11 and cfg[3] % 12 == 0:      # (tile 0 = the shipped route: convolutions on the loader-wave kernels)
        assert "gemm8p_kernel" in T.LAST_FUSED_KERNEL, T.LAST_FUSED_KERNEL    # (the bit-mask launches behind it are staged-epilogue only)


@pytest.mark.parametrize("tile", [10, 11])
@pytest.mark.parametrize("M,N", [(512, 192), (1024, 384), (300, 768)])
def test_conv_forward_eval_epilogue(tile, M, N):
    """eval-mode BatchNorm folded into a (1x1) convolution launch: C = relu(acc * colscale[n] + bias[n] + residual), one rounding"""
    ops = T._ops()
    Ci = 64
    Bn, Wd = 1, M
    geom = ops.ConvGeom(Bn, 1, Wd, Ci, N, 1, (1, 1), 0)
    A, B = T._ints((M, Ci), lo=-2, hi=3, seed=4), T._ints((N, Ci), lo=-2, hi=3, seed=5)
    g = torch.Generator().manual_seed(6)
    scale = torch.tensor([0.25, 0.5, 1.0, -0.5])[torch.randint(0, 4, (N,), generator=g)].double()
    bias = torch.randint(-8, 9, (N,), generator=g).double()
    res = torch.randint(-16, 17, (M, N), generator=g).double()
    pre = (A @ B.t()) * scale + bias
    for with_res, relu in ((False, False), (False, True), (True, True)):
        ref = pre + (res if with_res else 0.0)
        if relu:
            ref = ref.clamp_min(0.0)
        c = torch.empty(M, N, dtype=BF, device="cuda")
        ops.gemm(A.to(BF).cuda(), B.to(BF).cuda(), c, dtype=BF, M=M, N=N, K=Ci, lda=Ci, ldb=Ci, ldc=N, gather=ops.GATHER_CONV_FWD,
                 geom=geom, Cpad=Ci, colscale=scale.float().cuda(), bias=bias.float().cuda(),
                 residual=res.to(BF).cuda() if with_res else None, act=3 if relu else 0, tile=tile)
        assert "gemm8p_kernel" in _last_kernel(), _last_kernel()
        assert torch.equal(c.double().cpu(), ref.to(BF).double()), (with_res, relu)


def test_older_family_still_exact(force_tile):
    """tile 3 / 4: the one-barrier-per-k-tile LDS-DMA kernels (still the path of MN-major operands and float32 outputs)"""
    for tile in (3, 4):
        force_tile(tile)
        T.test_nt_plain_exact(BF, 512, 3072, 768)
        assert "gemm_dma_kernel" in _last_kernel()
        T.test_nt_epilogue(BF)


# --------------------------------------------------------------------------------------------------------------------
# halo-staged 3x3 stride-1 convolutions (csrc/gemm_halo_impl.h): forward (+ BatchNorm column sums) and dgrad against
# torch's float64 convolution on integer data, at shapes that exercise every edge of the halo bookkeeping
# --------------------------------------------------------------------------------------------------------------------
HALO = [  # B, H, W, Ci, Co
    (2, 4, 256, 192, 192),     # one tile per image row, 3 channel chunks, 192-column tile
    (1, 1, 256, 64, 128),      # H = 1: both neighbour rows outside the image; 128-column tile
    (3, 2, 512, 384, 384),     # two tiles per row (halo columns inside the image), two N tiles
    (2, 3, 256, 96, 256),      # Ci = 96 -> Cpad 128: the last chunk is half padding; Co = 256 -> 128-column tiles
    (1, 2, 768, 128, 64),      # three tiles per row, Co = 64: 64-column route (generic kernel) must still be right
]
HALO_STRIDED = [  # B, H, W, Ci, Co: stride (2, 1), the first conv of layer 1 (forward and weight gradient on the halo kernels)
    (2, 8, 256, 192, 192),
    (3, 5, 256, 64, 128),      # odd H: the last output row's third kernel row is padding
]


@pytest.mark.parametrize("cfg", HALO_STRIDED)
def test_halo_conv_row_stride_forward_wgrad_exact(cfg):
    ops = T._ops()
    Bn, Hh, Ww, Ci, Co = cfg
    x = T._ints((Bn, Ci, Hh, Ww), -2, 3, seed=80)
    w = T._ints((Co, Ci, 3, 3), -2, 3, seed=81).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=(2, 1), padding=1)
    dy = T._ints(tuple(y.shape), -2, 3, seed=82)
    y.backward(dy)
    geom = ops.ConvGeom(Bn, Hh, Ww, Ci, Co, 3, (2, 1), 1)
    M = Bn * geom.Ho * geom.Wo
    cpi = ops.cpad(Ci, BF)
    xd = x.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    wf = T._pack_fwd(w.detach(), cpi).to(BF).cuda()
    y_nhwc = y.detach().permute(0, 2, 3, 1)
    for tile in (12, 5):
        yd = torch.full((Bn, geom.Ho, geom.Wo, Co), 9.0, dtype=BF, device="cuda")
        nmt = ops.gemm_num_mtiles(M, Co, BF, gather=ops.GATHER_CONV_FWD)
        cs = torch.zeros(nmt, 2, Co, dtype=torch.float32, device="cuda")
        ops.gemm(xd, wf, yd, dtype=BF, M=M, N=Co, K=9 * cpi, lda=Ci, ldb=9 * cpi, ldc=Co, gather=ops.GATHER_CONV_FWD, geom=geom,
                 Cpad=cpi, colstats=cs, tile=tile)
        if tile == 12:
            assert "gemm_halo_kernel" in _last_kernel(), _last_kernel()
        assert torch.equal(yd.double().cpu(), y_nhwc.to(BF).double()), (tile, _last_kernel())
        assert torch.allclose(cs[:, 0].sum(0).double().cpu(), y_nhwc.reshape(-1, Co).sum(0), rtol=1e-6, atol=1e-3)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    want = T._pack_fwd(w.grad, cpi).permute(1, 2, 0)            # [taps][Cpad][Co]
    for tile, split_k in ((13, 8), (13, 3), (3, 8)):
        base = T._ints((9, cpi, Co), -5, 6, seed=83)
        dwp = base.float().cuda()
        ops.gemm(xd, dyd, dwp, dtype=BF, M=9 * cpi, N=Co, K=M, lda=Ci, ldb=Co, ldc=Co, a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR,
                 gather=ops.GATHER_CONV_WGRAD, geom=geom, Cpad=cpi, split_k=split_k, accumulate=True, c_f32=True, tile=tile)
        if tile == 13:
            assert "gemm_hwgrad" in _last_kernel(), _last_kernel()
        assert torch.equal(dwp.double().cpu(), base + want), (tile, split_k, float((dwp.double().cpu() - base - want).abs().max()))



HALO_COLSTRIDE = [  # B, H, W, Ci, Co, stride: conv1 of layer2.0 / layer3.0 (stride (2,2)) on odd / even pixel images (gemm_halo_fs2_kernel)
    (2, 8, 1024, 192, 384, (2, 2)),    # two tiles per output row (the odd image's first pixel inside the image for the second), 3 chunks, two 192-column tiles
    (3, 5, 512, 64, 128, (2, 2)),      # odd H: the last output row's third kernel row is padding; 128-column tile; one chunk
    (1, 1, 512, 96, 256, (2, 2)),      # H = 1: kernel rows 0 and 2 outside the image; Cpad 128 with a half-padded chunk; 128-column tiles
    (2, 3, 512, 128, 192, (1, 2)),     # row stride 1 with a column stride of 2
]


@pytest.mark.parametrize("cfg", HALO_COLSTRIDE)
def test_halo_conv_col_stride_forward_exact(cfg):
    """the column-strided 3x3 forward against torch's float64 convolution on integer data: raw output + BatchNorm column sums
    (training) and the eval-mode fold, the generic gather (tile 5) beside it"""
    ops = T._ops()
    Bn, Hh, Ww, Ci, Co, stride = cfg
    x = T._ints((Bn, Ci, Hh, Ww), -2, 3, seed=90)
    w = T._ints((Co, Ci, 3, 3), -2, 3, seed=91)
    y = F.conv2d(x, w, None, stride=stride, padding=1)
    geom = ops.ConvGeom(Bn, Hh, Ww, Ci, Co, 3, stride, 1)
    assert (geom.Ho, geom.Wo) == tuple(y.shape[2:])
    M = Bn * geom.Ho * geom.Wo
    cpi = ops.cpad(Ci, BF)
    xd = x.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    wf = T._pack_fwd(w, cpi).to(BF).cuda()
    y_nhwc = y.permute(0, 2, 3, 1)
    for tile in (12, 5):
        yd = torch.full((Bn, geom.Ho, geom.Wo, Co), 9.0, dtype=BF, device="cuda")
        nmt = ops.gemm_num_mtiles(M, Co, BF, gather=ops.GATHER_CONV_FWD)
        cs = torch.zeros(nmt, 2, Co, dtype=torch.float32, device="cuda")
        ops.gemm(xd, wf, yd, dtype=BF, M=M, N=Co, K=9 * cpi, lda=Ci, ldb=9 * cpi, ldc=Co, gather=ops.GATHER_CONV_FWD, geom=geom,
                 Cpad=cpi, colstats=cs, tile=tile)
        if tile == 12:
            assert "gemm_halo_fs2_kernel" in _last_kernel(), _last_kernel()
        assert torch.equal(yd.double().cpu(), y_nhwc.to(BF).double()), (tile, _last_kernel(), float((yd.double().cpu() - y_nhwc).abs().max()))
        assert torch.allclose(cs[:, 0].sum(0).double().cpu(), y_nhwc.reshape(-1, Co).sum(0), rtol=1e-6, atol=1e-3)
        assert torch.allclose(cs[:, 1].sum(0).double().cpu(), (y_nhwc.reshape(-1, Co) ** 2).sum(0), rtol=1e-6, atol=1e-3)
    # float32 C (the split-bf16 path's form of this convolution): exact sums, per-tile column sums
    yf = torch.full((Bn, geom.Ho, geom.Wo, Co), 9.0, dtype=torch.float32, device="cuda")
    nmt = ops.gemm_num_mtiles(M, Co, BF, gather=ops.GATHER_CONV_FWD)
    csf = torch.full((nmt, 2, Co), float("nan"), dtype=torch.float32, device="cuda")
    ops.gemm(xd, wf, yf, dtype=BF, M=M, N=Co, K=9 * cpi, lda=Ci, ldb=9 * cpi, ldc=Co, gather=ops.GATHER_CONV_FWD, geom=geom,
             Cpad=cpi, colstats=csf, c_f32=True, tile=12)
    assert "gemm_halo_fs2_kernel" in _last_kernel() and "f32" in _last_kernel(), _last_kernel()
    assert torch.equal(yf.double().cpu(), y_nhwc)
    yy = y_nhwc.reshape(-1, Co)
    assert torch.equal(csf.double().cpu(), torch.stack([torch.stack([yy[r0:r0 + 256].sum(0), (yy[r0:r0 + 256] ** 2).sum(0)]) for r0 in range(0, M, 256)]))
    g = torch.Generator().manual_seed(92)
    scale = torch.tensor([0.25, 0.5, 1.0, -0.5])[torch.randint(0, 4, (Co,), generator=g)].double()
    shift = torch.randint(-8, 9, (Co,), generator=g).double()
    ref = (y_nhwc * scale + shift).clamp_min(0.0)
    yd = torch.full((Bn, geom.Ho, geom.Wo, Co), 9.0, dtype=BF, device="cuda")
    ops.gemm(xd, wf, yd, dtype=BF, M=M, N=Co, K=9 * cpi, lda=Ci, ldb=9 * cpi, ldc=Co, gather=ops.GATHER_CONV_FWD, geom=geom,
             Cpad=cpi, colscale=scale.float().cuda(), bias=shift.float().cuda(), act=3, tile=12)
    assert "gemm_halo_fs2_kernel" in _last_kernel(), _last_kernel()
    assert torch.equal(yd.double().cpu(), ref.to(BF).double()), _last_kernel()


@pytest.mark.parametrize("cfg", [(2, 8, 256, 192, 384, (2, 2)),     # layer2.0-like: two 96-channel chunks, two 192-column tiles
                                 (3, 5, 128, 384, 192, (2, 2)),     # odd H, four chunks, one k-tile per output row
                                 (1, 3, 512, 192, 320, (1, 2)),     # row stride 1; Co = 320: a ragged second column tile
                                 (2, 4, 128, 96, 192, (2, 2))])     # Cpad 128 is not a multiple of 96: the generic kernel must serve it
def test_halo_conv_col_stride_wgrad_exact(cfg):
    """weight gradient of the column-strided 3x3 convolutions (gemm_hwgrad16_kernel<96, 192, 2>: odd / even pixel images of x) against
    torch's float64 autograd on integer data: slabs and atomics, XCD-grouped and ragged split factors, the generic gather beside it"""
    ops = T._ops()
    Bn, Hh, Ww, Ci, Co, stride = cfg
    x = T._ints((Bn, Ci, Hh, Ww), -2, 3, seed=95)
    w = T._ints((Co, Ci, 3, 3), -2, 3, seed=96).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=stride, padding=1)
    dy = T._ints(tuple(y.shape), -2, 3, seed=97)
    y.backward(dy)
    geom = ops.ConvGeom(Bn, Hh, Ww, Ci, Co, 3, stride, 1)
    M = Bn * geom.Ho * geom.Wo
    cpi = ops.cpad(Ci, BF)
    xd = x.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    want = T._pack_fwd(w.grad, cpi).permute(1, 2, 0)            # [taps][Cpad][Co]
    for tile, split_k in ((13, 8), (13, 3), (13, 1), (3, 8)):
        base = T._ints((9, cpi, Co), -5, 6, seed=98)
        dwp = base.float().cuda()
        ops.gemm(xd, dyd, dwp, dtype=BF, M=9 * cpi, N=Co, K=M, lda=Ci, ldb=Co, ldc=Co, a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR,
                 gather=ops.GATHER_CONV_WGRAD, geom=geom, Cpad=cpi, split_k=split_k, accumulate=True, c_f32=True, tile=tile)
        if tile == 13 and cpi % 96 == 0:
            assert "gemm_hwgrad16_kernel<96, 192, 2>" in _last_kernel(), _last_kernel()
        assert torch.equal(dwp.double().cpu(), base + want), (tile, split_k, _last_kernel(), float((dwp.double().cpu() - base - want).abs().max()))


@pytest.mark.parametrize("cfg", HALO)
def test_halo_conv_forward_dgrad_exact(cfg):
    ops = T._ops()
    Bn, Hh, Ww, Ci, Co = cfg
    x = T._ints((Bn, Ci, Hh, Ww), -2, 3, seed=50).requires_grad_(True)
    w = T._ints((Co, Ci, 3, 3), -2, 3, seed=51)
    y = F.conv2d(x, w, None, stride=1, padding=1)
    dy = T._ints(tuple(y.shape), -2, 3, seed=52)
    y.backward(dy)
    geom = ops.ConvGeom(Bn, Hh, Ww, Ci, Co, 3, (1, 1), 1)
    M = Bn * Hh * Ww
    cpi, cpo = ops.cpad(Ci, BF), ops.cpad(Co, BF)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    wf = T._pack_fwd(w, cpi).to(BF).cuda()
    y_nhwc = y.detach().permute(0, 2, 3, 1)
    outs = {}
    for tile in (5, 12):         # 5: generic gather kernel, 12: halo kernel where eligible
        yd = torch.full((Bn, Hh, Ww, Co), 9.0, dtype=BF, device="cuda")
        nmt = ops.gemm_num_mtiles(M, Co, BF, gather=ops.GATHER_CONV_FWD)
        cs = torch.zeros(nmt, 2, Co, dtype=torch.float32, device="cuda")
        ops.gemm(xd, wf, yd, dtype=BF, M=M, N=Co, K=9 * cpi, lda=Ci, ldb=9 * cpi, ldc=Co, gather=ops.GATHER_CONV_FWD, geom=geom,
                 Cpad=cpi, colstats=cs, tile=tile)
        outs[tile] = _last_kernel()
        assert torch.equal(yd.double().cpu(), y_nhwc.to(BF).double()), (tile, _last_kernel())
        assert torch.allclose(cs[:, 0].sum(0).double().cpu(), y_nhwc.reshape(-1, Co).sum(0), rtol=1e-6, atol=1e-3)
        assert torch.allclose(cs[:, 1].sum(0).double().cpu(), (y_nhwc.reshape(-1, Co) ** 2).sum(0), rtol=1e-6, atol=1e-3)
        dyd = dy.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
        wd = T._pack_dgrad(w, cpo).to(BF).cuda()
        dxd = torch.full((Bn, Hh, Ww, Ci), 9.0, dtype=BF, device="cuda")
        ops.gemm(dyd, wd, dxd, dtype=BF, M=M, N=Ci, K=9 * cpo, lda=Co, ldb=9 * cpo, ldc=Ci, gather=ops.GATHER_CONV_DGRAD, geom=geom,
                 Cpad=cpo, tile=tile)
        assert torch.equal(dxd.double().cpu(), x.grad.permute(0, 2, 3, 1).to(BF).double()), (tile, _last_kernel())
    if Ci >= 96:       # the dgrad's N = Ci: 192- or 128-column tiles -> the halo kernel must have run under tile 12
        assert "gemm_halo_kernel" in outs[12] or "gemm_halo_kernel" in _last_kernel(), (outs, _last_kernel())


@pytest.mark.parametrize("cfg", [(2, 4, 256, 192, 192), (3, 2, 512, 384, 384), (1, 1, 256, 64, 128), (2, 3, 256, 96, 256)])
def test_halo_conv_float32_output_exact(cfg):
    """the halo kernels with float32 C (round 5: the convolutions of the split-bf16 parity path): forward + per-tile column sums,
    dgrad with and without a float32 residual, against torch's float64 convolution on integer data -- exact, no rounding of the
    result -- and the generic gather (tile 5) beside them"""
    ops = T._ops()
    Bn, Hh, Ww, Ci, Co = cfg
    x = T._ints((Bn, Ci, Hh, Ww), -2, 3, seed=150).requires_grad_(True)
    w = T._ints((Co, Ci, 3, 3), -2, 3, seed=151)
    y = F.conv2d(x, w, None, stride=1, padding=1)
    dy = T._ints(tuple(y.shape), -2, 3, seed=152)
    y.backward(dy)
    geom = ops.ConvGeom(Bn, Hh, Ww, Ci, Co, 3, (1, 1), 1)
    M = Bn * Hh * Ww
    cpi, cpo = ops.cpad(Ci, BF), ops.cpad(Co, BF)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    wf = T._pack_fwd(w, cpi).to(BF).cuda()
    y_nhwc = y.detach().permute(0, 2, 3, 1)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    wd = T._pack_dgrad(w, cpo).to(BF).cuda()
    res = T._ints((Bn, Hh, Ww, Ci), -50, 51, seed=153)
    dx_ref = x.grad.permute(0, 2, 3, 1)
    for tile in (12, 5):
        yd = torch.full((Bn, Hh, Ww, Co), 9.0, dtype=torch.float32, device="cuda")
        nmt = ops.gemm_num_mtiles(M, Co, BF, gather=ops.GATHER_CONV_FWD)
        cs = torch.full((nmt, 2, Co), float("nan"), dtype=torch.float32, device="cuda")
        ops.gemm(xd, wf, yd, dtype=BF, M=M, N=Co, K=9 * cpi, lda=Ci, ldb=9 * cpi, ldc=Co, gather=ops.GATHER_CONV_FWD, geom=geom,
                 Cpad=cpi, colstats=cs, c_f32=True, tile=tile)
        if tile == 12 and Co >= 96:
            assert "gemm_halo_kernel" in _last_kernel() and "f32" in _last_kernel(), _last_kernel()
        assert torch.equal(yd.double().cpu(), y_nhwc), (tile, _last_kernel(), float((yd.double().cpu() - y_nhwc).abs().max()))
        yy = y_nhwc.reshape(-1, Co)
        want = torch.stack([torch.stack([yy[r0:r0 + 256].sum(0), (yy[r0:r0 + 256] ** 2).sum(0)]) for r0 in range(0, M, 256)])
        assert torch.equal(cs.double().cpu(), want), (tile, _last_kernel())
        for with_res in (False, True):
            dxd = torch.full((Bn, Hh, Ww, Ci), 9.0, dtype=torch.float32, device="cuda")
            ops.gemm(dyd, wd, dxd, dtype=BF, M=M, N=Ci, K=9 * cpo, lda=Co, ldb=9 * cpo, ldc=Ci, gather=ops.GATHER_CONV_DGRAD, geom=geom,
                     Cpad=cpo, c_f32=True, residual=res.float().cuda() if with_res else None, tile=tile)
            if tile == 12 and Ci >= 96:
                assert "gemm_halo_kernel" in _last_kernel() and "f32" in _last_kernel(), _last_kernel()
            assert torch.equal(dxd.double().cpu(), dx_ref + (res if with_res else 0.0)), (tile, with_res, _last_kernel())


@pytest.mark.parametrize("cfg", [(2, 2, 256, 192, 192), (1, 3, 512, 128, 384)])
def test_halo_conv_forward_eval_fold_exact(cfg):
    """eval-mode BatchNorm (+ residual + ReLU) folded into a halo-staged 3x3 convolution: relu(conv * scale + shift + res),
    one rounding, against float64 on integer data with power-of-two scales"""
    ops = T._ops()
    Bn, Hh, Ww, Ci, Co = cfg
    x = T._ints((Bn, Ci, Hh, Ww), -2, 3, seed=60)
    w = T._ints((Co, Ci, 3, 3), -1, 2, seed=61)
    y = F.conv2d(x, w, None, stride=1, padding=1).permute(0, 2, 3, 1)
    g = torch.Generator().manual_seed(62)
    scale = torch.tensor([0.25, 0.5, 1.0, -0.5])[torch.randint(0, 4, (Co,), generator=g)].double()
    shift = torch.randint(-8, 9, (Co,), generator=g).double()
    res = torch.randint(-16, 17, tuple(y.shape), generator=g).double()
    geom = ops.ConvGeom(Bn, Hh, Ww, Ci, Co, 3, (1, 1), 1)
    M, cpi = Bn * Hh * Ww, ops.cpad(Ci, BF)
    xd = x.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    wf = T._pack_fwd(w, cpi).to(BF).cuda()
    for with_res, relu in ((False, False), (False, True), (True, True)):
        ref = y * scale + shift + (res if with_res else 0.0)
        if relu:
            ref = ref.clamp_min(0.0)
        for tile in (12, 5):
            yd = torch.full((Bn, Hh, Ww, Co), 9.0, dtype=BF, device="cuda")
            ops.gemm(xd, wf, yd, dtype=BF, M=M, N=Co, K=9 * cpi, lda=Ci, ldb=9 * cpi, ldc=Co, gather=ops.GATHER_CONV_FWD, geom=geom,
                     Cpad=cpi, colscale=scale.float().cuda(), bias=shift.float().cuda(),
                     residual=res.to(BF).cuda() if with_res else None, act=3 if relu else 0, tile=tile)
            if tile == 12:
                assert "gemm_halo_kernel" in _last_kernel(), _last_kernel()
            assert torch.equal(yd.double().cpu(), ref.to(BF).double()), (with_res, relu, tile, _last_kernel())


# --------------------------------------------------------------------------------------------------------------------
# first block of a stage (resnet18.py:33-37,59-63): the input gradient of the 1x1 stride-s downsample conv formed inside
# the class-(0,0) launch of the strided 3x3 conv's parity-class dgrad (HtrvtGemmDesc.A2: one more tap)
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [(2, 8, 128, 192, 384, (2, 2)), (2, 16, 128, 192, 192, (2, 1)), (3, 4, 192, 384, 768, (2, 2)),
                                 (2, 8, 144, 192, 384, (2, 2))])
@pytest.mark.parametrize("fused_epilogue", [False, True])
def test_strided_dgrad_with_downsample_gradient_as_extra_tap(cfg, fused_epilogue):
    import htrvt_amd  # noqa: F401
    from htrvt_amd.engine import Engine, ModelShape
    ops = T._ops()
    Bn, Hi, Wi, Ci, Co, stride = cfg
    eng = Engine(ModelShape(80, (64, 512), 64, 2, 2), BF, "cuda")
    g3 = ops.ConvGeom(Bn, Hi, Wi, Ci, Co, 3, stride, 1)
    gd = ops.ConvGeom(Bn, Hi, Wi, Ci, Co, 1, stride, 0)
    assert (g3.Ho, g3.Wo) == (gd.Ho, gd.Wo)
    w3 = T._sparse_ints((Co, Ci, 3, 3), 60)
    wd = T._sparse_ints((Co, Ci, 1, 1), 61)
    dy3 = T._sparse_ints((Bn, Co, g3.Ho, g3.Wo), 62)
    dyd = T._sparse_ints((Bn, Co, g3.Ho, g3.Wo), 63)
    dx = (torch.nn.grad.conv2d_input((Bn, Ci, Hi, Wi), w3, dy3, stride=stride, padding=1)
          + torch.nn.grad.conv2d_input((Bn, Ci, Hi, Wi), wd, dyd, stride=stride, padding=0)).permute(0, 2, 3, 1)
    pair = torch.empty(2, Bn, g3.Ho, g3.Wo, Co, dtype=BF, device="cuda")
    pair[0].copy_(dy3.permute(0, 2, 3, 1).to(BF))
    pair[1].copy_(dyd.permute(0, 2, 3, 1).to(BF))
    wj = eng._conv_w_joint_dgrad("t3", w3.float().cuda(), "td", wd.float().cuda())
    assert wj.shape == (Ci, 10, ops.cpad(Co, BF))
    kw, want = {}, dx
    if fused_epilogue:
        relu_src = T._ints((Bn, Hi, Wi, Ci), -1, 2, seed=64)
        bnx = T._ints((Bn, Hi, Wi, Ci), -4, 5, seed=65)
        gen = torch.Generator().manual_seed(66)
        mean = torch.randint(-2, 3, (Ci,), generator=gen).double()
        rstd = torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (Ci,), generator=gen)].double()
        rows = eng.dgrad_tiles(g3)
        part = torch.full((rows, 2, Ci), float("nan"), dtype=torch.float32, device="cuda")
        kw = dict(relu_src=relu_src.to(BF).cuda(), bnb=[(bnx.to(BF).cuda(), mean.float().cuda(), rstd.float().cuda(), part)])
        want = dx * (relu_src > 0)
    assert want.abs().max() < 256
    out = eng.conv_dgrad(pair[0], wj, g3, extra=pair[1], **kw)
    assert torch.equal(out.double().cpu(), want), float((out.double().cpu() - want).abs().max())
    if fused_epilogue:      # the partial sums of all class launches together = the sums over the whole gradient
        got = part.double().cpu().sum(0)
        xhat = (bnx - mean) * rstd
        assert torch.equal(got[0], want.reshape(-1, Ci).sum(0))
        assert torch.equal(got[1], (want * xhat).reshape(-1, Ci).sum(0))


# --------------------------------------------------------------------------------------------------------------------
# halo-staged weight gradient of the 3x3 stride-1 convolutions (csrc/gemm_hwgrad_impl.h) on integer data: atomics and
# slabs, XCD-grouped and plain split factors, channel chunks of 128 and 64, 192- and 128-column tiles, ragged pixel ranges
# --------------------------------------------------------------------------------------------------------------------
HWGRAD = [  # B, H, W, Ci, Co, split_k
    (2, 4, 256, 192, 192, 8),      # Cpad 192 -> chunks of 64, three of them; XCD-grouped split
    (2, 3, 128, 384, 384, 3),      # chunks of 128, two N tiles, plain split
    (1, 2, 256, 768, 768, 1),      # no split at all
    (3, 2, 64, 96, 256, 5),        # Ci = 96 -> Cpad 128 with a quarter padding; Co = 256 -> 128-column tiles; W = 64
    (2, 8, 192, 64, 64, 16),       # Ci = Co = 64: one chunk, half-empty 128-column tile
    (1, 3, 192, 160, 192, 3),      # Ci = 160 -> Cpad 192: 96-channel units (16x16x32 kernel) with padded channels, plain split, W = 192
]


@pytest.mark.parametrize("cfg", HWGRAD)
@pytest.mark.parametrize("slabs", [False, True])
def test_halo_conv_wgrad_exact(cfg, slabs):
    ops = T._ops()
    Bn, Hh, Ww, Ci, Co, split_k = cfg
    x = T._ints((Bn, Ci, Hh, Ww), -2, 3, seed=70)
    w = T._ints((Co, Ci, 3, 3), -1, 2, seed=71).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=1, padding=1)
    dy = T._ints(tuple(y.shape), -2, 3, seed=72)
    y.backward(dy)
    geom = ops.ConvGeom(Bn, Hh, Ww, Ci, Co, 3, (1, 1), 1)
    M = Bn * Hh * Ww
    cpi = ops.cpad(Ci, BF)
    xd = x.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
    want = T._pack_fwd(w.grad, cpi).permute(1, 2, 0)            # [taps][Cpad][Co]
    tiles = (13, 3) + ((18,) if cpi % 128 else ())              # 13: halo kernel (Cpad 192: the 16x16x32 form), 3: the generic MN-major gather, 18: the paired 32x32x16 form
    for tile in tiles:
        base = T._ints((9, cpi, Co), -5, 6, seed=73)
        dwp = base.float().cuda()
        ws = torch.empty(max(split_k, 1), 9 * cpi, Co, dtype=torch.float32, device="cuda") if (slabs and split_k > 1) else None
        ops.gemm(xd, dyd, dwp, dtype=BF, M=9 * cpi, N=Co, K=M, lda=Ci, ldb=Co, ldc=Co, a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR,
                 gather=ops.GATHER_CONV_WGRAD, geom=geom, Cpad=cpi, split_k=split_k, accumulate=True, c_f32=True, splitk_ws=ws, tile=tile)
        if tile in (13, 18):
            assert "gemm_hwgrad" in _last_kernel(), _last_kernel()
        if tile == 13 and cpi % 128 and cpi % 96 == 0 and Co > 128:
            assert "gemm_hwgrad16_kernel" in _last_kernel(), _last_kernel()
        assert torch.equal(dwp.double().cpu(), base + want), (tile, float((dwp.double().cpu() - base - want).abs().max()))


@pytest.mark.parametrize("split_k", [2, 5, 7, 9, 13, 21, 42])
def test_wgrad_any_split_factor_exact(split_k):
    """round 5: split factors that are not multiples of 8 run on a 1-D grid with (range, tile) pairs consecutive per XCD
    (xcd_range_map, csrc/gemm_dma_impl.h) -- a bijection of workgroup ids for every (tiles, split) pair, also where the workgroup
    count is not a multiple of 8.  Conv weight gradient on integer data through the three split-K kernels (halo 32x32x16, halo
    16x16x32 with 96-channel units, generic MN-major gather), slabs and atomics, ranges that do not divide the pixel count."""
    ops = T._ops()
    for Bn, Hh, Ww, Ci, Co in ((3, 3, 192, 384, 192), (2, 5, 128, 192, 384)):      # 9 x 1 / 6 x 2 halo tiles; 1 728 / 1 280 pixels
        x = T._ints((Bn, Ci, Hh, Ww), -2, 3, seed=74)
        w = T._ints((Co, Ci, 3, 3), -1, 2, seed=75).requires_grad_(True)
        y = F.conv2d(x, w, None, stride=1, padding=1)
        dy = T._ints(tuple(y.shape), -2, 3, seed=76)
        y.backward(dy)
        geom = ops.ConvGeom(Bn, Hh, Ww, Ci, Co, 3, (1, 1), 1)
        M = Bn * Hh * Ww
        cpi = ops.cpad(Ci, BF)
        xd = x.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
        dyd = dy.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
        want = T._pack_fwd(w.grad, cpi).permute(1, 2, 0)
        for tile in (13, 3):
            for slabs in (False, True):
                base = T._ints((9, cpi, Co), -5, 6, seed=77)
                dwp = base.float().cuda()
                ws = torch.empty(split_k, 9 * cpi, Co, dtype=torch.float32, device="cuda") if slabs else None
                ops.gemm(xd, dyd, dwp, dtype=BF, M=9 * cpi, N=Co, K=M, lda=Ci, ldb=Co, ldc=Co, a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR,
                         gather=ops.GATHER_CONV_WGRAD, geom=geom, Cpad=cpi, split_k=split_k, accumulate=True, c_f32=True, splitk_ws=ws, tile=tile)
                assert torch.equal(dwp.double().cpu(), base + want), (split_k, tile, slabs, _last_kernel(), float((dwp.double().cpu() - base - want).abs().max()))


# ---------------------------------------------------------------------------------------------------------------------
# Persistent form (csrc/gemm8pp_impl.h): one workgroup per CU walks its tiles, the epilogue of a tile is folded into the
# first k-tile of the next.  Shapes with MORE tiles than CUs (several tiles per workgroup: the folded flush), exactly as
# many (every tile is a workgroup's last: the trailing flush), row / column tails, both tile widths, padded leading dims.
# ---------------------------------------------------------------------------------------------------------------------
def _ncu():
    return torch.cuda.get_device_properties(0).multi_processor_count & ~7


@pytest.mark.parametrize("M,N,K,tile", [(8192, 4608, 384, 14),     # 32 x 18 tiles of 256 x 256
                                        (4096, 4096, 256, 14),     # 16 x 16 = 256 tiles: one per CU
                                        (8000, 2312, 512, 14),     # row tail (8000 = 31.25 tiles), column tail
                                        (16384, 768, 768, 15),     # 64 x 4 tiles of 256 x 192 (the proj / fc2 width)
                                        (10000, 1536, 256, 15),
                                        (8192, 4608, 384, 15)])
def test_persistent_plain_exact(M, N, K, tile):
    ops = T._ops()
    A, B = T._ints((M, K), seed=11), T._ints((N, K), seed=12)
    bias = T._ints((N,), lo=-20, hi=21, seed=13)
    ref = (A @ B.t() + bias).to(BF).double()
    a, b = A.to(BF).cuda(), B.to(BF).cuda()
    for with_bias in (True, False):
        c = torch.full((M, N + 8), 7.0, dtype=BF, device="cuda")
        ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N + 8, bias=bias.float().cuda() if with_bias else None, tile=tile)
        k = _last_kernel()
        assert "gemm8pp_kernel" in k and f"Cfg<{256 if tile == 14 else 192}" in k, k
        got = c.double().cpu()
        want = ref if with_bias else (A @ B.t()).to(BF).double()
        assert torch.equal(got[:, :N], want), (with_bias, int((got[:, :N] != want).sum()))
        assert (got[:, N:] == 7.0).all()
    # the auto route (its own choice of kernel and width; the model's shapes are pinned in test_persistent_repeated_launches...)
    c = torch.empty((M, N), dtype=BF, device="cuda")
    ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias.float().cuda())
    assert torch.equal(c.double().cpu(), ref), _last_kernel()


@pytest.mark.parametrize("M,N,K,tiles", [(8192, 3072, 768, (14, 10)), (16384, 768, 256, (15, 11)), (8192, 3072, 768, (15, 11))])
def test_persistent_gelu_epilogue(M, N, K, tiles):
    """bias + exact-erf GELU + saved pre-activation (fc1 forward) through the folded flush: bit-identical to the
    one-tile-per-workgroup kernel (same arithmetic, same rounding points), and both against float64"""
    ops = T._ops()
    g = torch.Generator().manual_seed(5)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    a, b = A.to(BF).cuda(), B.to(BF).cuda()
    outs = {}
    for tile in tiles:
        c = torch.empty(M, N, dtype=BF, device="cuda")
        pr = torch.empty(M, N, dtype=BF, device="cuda")
        ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, alpha=0.5, act=1, bias=bias.cuda(), preact=pr, tile=tile)
        assert ("gemm8pp_kernel" if tile >= 13 else "gemm8p_kernel<") in _last_kernel(), _last_kernel()
        outs[tile] = (c, pr)
    assert torch.equal(outs[tiles[0]][0], outs[tiles[1]][0]) and torch.equal(outs[tiles[0]][1], outs[tiles[1]][1])
    pre = (a.double() @ b.double().t()) * 0.5 + bias.double().cuda()
    pq = outs[tiles[0]][1].double()
    assert (pq - pre).abs().max() <= 2.0 ** -8 * pre.abs().max()
    assert (outs[tiles[0]][0].double() - F.gelu(pq)).abs().max() <= 2.0 ** -8 * pq.abs().max() + 2e-6
    # without a saved pre-activation
    c2 = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm(a, b, c2, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, alpha=0.5, act=1, bias=bias.cuda(), tile=tiles[0])
    assert "gemm8pp_kernel" in _last_kernel()
    assert torch.equal(c2, outs[tiles[0]][0])


@pytest.mark.parametrize("M,N,K", [(32768, 768, 768), (32768, 768, 3072), (16384, 768, 256), (16500, 780, 384)])
def test_persistent_residual_epilogue(M, N, K):
    """bias + residual (proj / fc2 forward) on the persistent kernel, the residual through inline-asm register loads
    with counted waits (csrc/gemm8pp_impl.h, struct Side): exact on integer data (M / N tails, padded ldc), and -- the race
    screen -- ten launches on random data bit-identical to the one-tile-per-workgroup kernel"""
    ops = T._ops()
    A, B = T._ints((M, K), lo=-2, hi=3, seed=31), T._ints((N, K), lo=-2, hi=3, seed=32)
    bias = T._ints((N,), lo=-20, hi=21, seed=33)
    res = T._ints((M, N + 8), lo=-30, hi=31, seed=34)
    ref = (A @ B.t() + bias + res[:, :N]).to(BF).double()
    a, b, r = A.to(BF).cuda(), B.to(BF).cuda(), res.to(BF).cuda()
    c = torch.full((M, N + 8), 7.0, dtype=BF, device="cuda")
    ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N + 8, bias=bias.float().cuda(), residual=r)
    assert "gemm8pp_kernel<Cfg<192, 4, 2>, 1>" in _last_kernel(), _last_kernel()
    got = c.double().cpu()
    assert torch.equal(got[:, :N], ref), int((got[:, :N] != ref).sum())
    assert (got[:, N:] == 7.0).all()
    g = torch.Generator().manual_seed(35)
    a = torch.randn(M, K, generator=g).to(BF).cuda()
    b = (torch.randn(N, K, generator=g) * 0.05).to(BF).cuda()
    r = torch.randn(M, N, generator=g).to(BF).cuda()
    biasf = torch.randn(N, generator=g).cuda()
    want = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm(a, b, want, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=biasf, residual=r, tile=11)
    assert "gemm8p_kernel<" in _last_kernel()
    for it in range(10):
        c = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
        ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=biasf, residual=r)
        assert "gemm8pp_kernel" in _last_kernel()
        assert torch.equal(c, want), (it, int((c != want).sum()))


def test_persistent_repeated_launches_are_bit_identical():
    """race screen: the folded flush reads accumulators the next phase overwrites and a bias slot the next tile's DMA
    refills; 20 launches of the model's qkv shape on random data must agree bit for bit with the first and with the
    one-tile-per-workgroup kernel"""
    ops = T._ops()
    M, N, K = 32768, 2304, 768
    g = torch.Generator().manual_seed(9)
    a = (torch.randn(M, K, generator=g)).to(BF).cuda()
    b = (torch.randn(N, K, generator=g) * 0.05).to(BF).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm(a, b, ref, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, tile=9)
    assert "gemm8p_kernel<" in _last_kernel()
    for it in range(20):
        c = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
        ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias)
        assert "gemm8pp_kernel" in _last_kernel()
        assert torch.equal(c, ref), (it, int((c != ref).sum()))


# ---------------------------------------------------------------------------------------------------------------------
# MN-major x MN-major operands on the 8-phase schedule (csrc/gemm8pt_impl.h): C = A^T B, float32, the Linear weight
# gradients dW = dy^T x.  Exact integer data: row / column / K tails, padded leading dimensions, split-K into slabs.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,split,pad", [(768, 768, 4096, 1, 0), (3072, 768, 8192, 4, 0), (776, 264, 1000, 1, 8),
                                             (2304, 768, 32768, 9, 0), (520, 1032, 2050, 3, 16), (256, 256, 256, 1, 0)])
def test_mnmajor_8phase_exact(M, N, K, split, pad):
    ops = T._ops()
    A, B = T._ints((K, M + pad), lo=-2, hi=3, seed=21), T._ints((K, N + pad), lo=-2, hi=3, seed=22)
    ref = A[:, :M].t() @ B[:, :N]
    a, b = A.to(BF).cuda(), B.to(BF).cuda()
    base = T._ints((M, N + 4), lo=-5, hi=6, seed=23).float().cuda()
    for tile in (16, 3):       # this kernel, and the older one-barrier-per-k-tile kernel beside it
        c = base.clone()
        ws = torch.empty(split, M, N, dtype=torch.float32, device="cuda") if split > 1 else None
        if split > 1:
            ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=M + pad, ldb=N + pad, ldc=N + 4, a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR,
                     split_k=split, accumulate=True, c_f32=True, splitk_ws=ws, tile=tile)
            want = base.double().cpu()
            want[:, :N] += ref
        else:
            ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=M + pad, ldb=N + pad, ldc=N + 4, a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR,
                     c_f32=True, tile=tile)
            want = base.double().cpu()
            want[:, :N] = ref
        k = _last_kernel()
        assert ("gemm8pt_kernel" in k) == (tile == 16), k
        assert torch.equal(c.double().cpu(), want), (tile, int((c.double().cpu() != want).sum()))


def test_mnmajor_8phase_repeated_launches_bit_identical():
    """race screen on the model's fc1 weight-gradient shape, random data, slab split-K: 10 launches agree bit for bit and
    match float64 to float32 rounding"""
    ops = T._ops()
    M, N, K, split = 3072, 768, 32768, 7
    g = torch.Generator().manual_seed(4)
    a = torch.randn(K, M, generator=g).to(BF).cuda()
    b = torch.randn(K, N, generator=g).to(BF).cuda()
    outs = []
    for it in range(10):
        c = torch.zeros(M, N, dtype=torch.float32, device="cuda")
        ws = torch.full((split, M, N), float("nan"), dtype=torch.float32, device="cuda")
        ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, a_layout=ops.MNMAJOR, b_layout=ops.MNMAJOR,
                 split_k=split, accumulate=True, c_f32=True, splitk_ws=ws)
        assert "gemm8pt_kernel" in _last_kernel(), _last_kernel()
        outs.append(c)
    for c in outs[1:]:
        assert torch.equal(c, outs[0])
    ref = a[:, :256].double().t() @ b.double()
    assert (outs[0][:256].double() - ref).abs().max() <= 1e-5 * ref.abs().max()


@pytest.mark.parametrize("tile", [10, 11])
@pytest.mark.parametrize("M,N,K", [(512, 768, 2304), (1000, 264, 200), (4096, 3072, 192)])
def test_plain_float32_output_exact(tile, M, N, K):
    """float32 C straight from the accumulators (+ bias): the Linear products of the split-bf16 parity path"""
    ops = T._ops()
    A, B = T._ints((M, K), seed=31), T._ints((N, K), seed=32)
    bias = T._ints((N,), lo=-9, hi=10, seed=33)
    a, b = A.to(BF).cuda(), B.to(BF).cuda()
    c = torch.full((M, N + 4), 3.0, dtype=torch.float32, device="cuda")
    ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N + 4, bias=bias.float().cuda(), c_f32=True, tile=tile)
    assert "gemm8p_kernel" in _last_kernel() and ", 128>" in _last_kernel(), _last_kernel()
    got = c.double().cpu()
    assert torch.equal(got[:, :N], A @ B.t() + bias)
    assert (got[:, N:] == 3.0).all()
