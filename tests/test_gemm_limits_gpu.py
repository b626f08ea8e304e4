"""The 2 GiB boundary of htrvt_gemm (include/htrvt.h; SURVEY 8(b): "return a negative code, never throw / abort").

The LDS-DMA kernel families address every operand through a 2 GiB buffer descriptor.  An operand beyond that is either
served by the register-staged kernel (64-bit addressing: plain contractions, tile = 0) or REFUSED before anything is
launched -- the forms that exist in the LDS-DMA families only: per-tile column sums sized for 256-row tiles, fused backward
epilogues, parity-class / merged strided dgrad, A2, and any explicit family selector.  Round 4's memory fault was a missing
refusal of the first kind (csrc/gemm.hip, launch_main).  The refusing descriptors here point at SMALL real buffers and only
claim large extents: the checks precede the launch, so nothing may touch them."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _env():
    import htrvt_amd  # noqa: F401
    from htrvt_amd import ops
    from htrvt_amd._lib import lib
    return ops, lib


def _sentinel(ops, lib):
    """a small valid launch: htrvt_last_kernel() then names ITS kernel until another launch happens"""
    a = torch.ones(256, 64, dtype=BF, device="cuda")
    c = torch.empty(256, 256, dtype=BF, device="cuda")
    ops.gemm(a, a, c, dtype=BF, M=256, N=256, K=64, lda=64, ldb=64, ldc=256)
    torch.cuda.synchronize()
    return lib.htrvt_last_kernel().decode()


def _refused(ops, lib, what, **kw):
    before = _sentinel(ops, lib)
    with pytest.raises(RuntimeError) as ei:
        ops.gemm(**kw)
    msg = str(ei.value)
    assert "htrvt_gemm" in msg and len(msg) > 30, (what, msg)
    assert lib.htrvt_last_kernel().decode() == before, (what, "a kernel was launched")
    torch.cuda.synchronize()        # and nothing faulted
    return msg


def test_plain_operand_above_2gib_is_served_by_the_register_staged_kernel():
    """tile = 0, a REAL 2.2 GiB A operand: the auto route declines the LDS-DMA families and the 64-bit kernel computes it"""
    ops, lib = _env()
    M, N, K = 1_100_000, 64, 1024                 # A: 2.25 GB
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randint(-2, 3, (M, K), generator=g, device="cuda", dtype=torch.int8).to(BF)
    b = torch.randint(-2, 3, (N, K), generator=g, device="cuda", dtype=torch.int8).to(BF)
    c = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm(a, b, c, dtype=BF, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    assert lib.htrvt_last_kernel().decode().startswith("gemm_kernel<"), lib.htrvt_last_kernel().decode()
    rows = torch.tensor([0, 1, 255, 256, 524287, 524288, 1048575, 1048576, M - 1], device="cuda")     # both sides of the 2^31-byte row
    want = (a[rows].float() @ b.float().t()).to(BF)
    assert torch.equal(c[rows], want)


@pytest.mark.parametrize("tile", [9, 12, 13, 4])
def test_explicit_family_selector_refuses_operands_above_2gib(tile):
    ops, lib = _env()
    a = torch.ones(4096, 1024, dtype=BF, device="cuda")
    b = torch.ones(64, 1024, dtype=BF, device="cuda")
    c = torch.empty(4096, 64, dtype=BF, device="cuda")
    msg = _refused(ops, lib, f"tile {tile}", A=a, B=b, Cout=c, dtype=BF, M=1_100_000, N=64, K=1024, lda=1024, ldb=1024, ldc=64, tile=tile)
    assert "2 GiB" in msg


def _conv_desc(ops, B, Hi, Wi, Ci, Co, k, stride, pad):
    g = ops.ConvGeom(B, Hi, Wi, Ci, Co, k, stride, pad)
    return g, ops.cpad(Ci, BF), ops.cpad(Co, BF)


def test_column_sums_refuse_operands_above_2gib():
    """conv forward with per-tile BatchNorm sums whose gathered tensor claims 2.4 GiB: the 128-row fallback would write twice
    the rows htrvt_gemm_num_mtiles sized -- refused (the split-bf16 engine slices the batch instead, engine.conv_fwd)"""
    ops, lib = _env()
    B, Hi, Wi, Ci, Co = 128, 8, 1024, 1152, 192                    # 128 x 8 x 1024 x 1152 x 2 B = 2.4 GB claimed
    g, cpi, _ = _conv_desc(ops, B, Hi, Wi, Ci, Co, 3, (1, 1), 1)
    x = torch.ones(1 << 20, dtype=BF, device="cuda")
    w = torch.ones(Co, 9, cpi, dtype=BF, device="cuda")
    y = torch.empty(1 << 20, dtype=BF, device="cuda")
    M = B * g.Ho * g.Wo
    cs = torch.empty(ops.gemm_num_mtiles(M, Co, BF, gather=ops.GATHER_CONV_FWD), 2, Co, dtype=torch.float32, device="cuda")
    msg = _refused(ops, lib, "colstats", A=x, B=w, Cout=y, dtype=BF, M=M, N=Co, K=9 * cpi, lda=Ci, ldb=9 * cpi, ldc=Co,
                   gather=ops.GATHER_CONV_FWD, geom=g, Cpad=cpi, colstats=cs)
    assert "2 GiB" in msg or "split the batch" in msg


@pytest.mark.parametrize("form", ["fused", "class", "merged", "a2"])
def test_dgrad_forms_refuse_operands_above_2gib(form):
    ops, lib = _env()
    stride = (1, 1) if form == "fused" else (2, 2)
    B, Hi, Wi, Ci, Co = 128, 16, 1024, 192, 2304 if form == "fused" else 4608     # dY claims 128 x Ho x Wo x Co x 2 B > 2 GiB
    g, _, cpo = _conv_desc(ops, B, Hi, Wi, Ci, Co, 3, stride, 1)
    assert B * g.Ho * g.Wo * Co * 2 >= 2 ** 31
    dy = torch.ones(1 << 20, dtype=BF, device="cuda")
    wd = torch.ones(Ci, 10, cpo, dtype=BF, device="cuda")
    dx = torch.empty(1 << 20, dtype=BF, device="cuda")
    side = torch.ones(1 << 20, dtype=BF, device="cuda")
    f1 = torch.ones(Ci, dtype=torch.float32, device="cuda")
    part = torch.empty(1 << 16, dtype=torch.float32, device="cuda")
    kw = dict(A=dy, B=wd, Cout=dx, dtype=BF, N=Ci, lda=Co, ldc=Ci, gather=ops.GATHER_CONV_DGRAD, geom=g, Cpad=cpo)
    if form == "fused":
        kw.update(M=B * Hi * Wi, K=9 * cpo, ldb=9 * cpo, relu_src=side, bnb=[(side, f1, f1, part)])
    elif form == "class":
        kw.update(M=B * (Hi // 2) * (Wi // 2), K=4 * cpo, ldb=9 * cpo, cls=(1, 1))
    elif form == "merged":
        kw.update(M=B * Hi * Wi, K=9 * cpo, ldb=9 * cpo, cls=(-2, -2))
    else:
        kw.update(M=B * (Hi // 2) * (Wi // 2), K=2 * cpo, ldb=10 * cpo, cls=(0, 0), a2=side)
    _refused(ops, lib, form, **kw)
