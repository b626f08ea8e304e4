"""Exact-integer checks (through the C ABI) of the layout / reduction helpers around the contraction kernels:
conv weight packing, weight-gradient unpacking, column sums (bias gradients, masked-token gradient)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    import htrvt_amd  # noqa: F401
    from htrvt_amd._lib import check, lib
    from htrvt_amd.ops import ptr, stream
    return lib, check, ptr, stream


@pytest.mark.parametrize("Co,Ci,taps", [(192, 192, 9), (384, 192, 1), (40, 24, 9), (768, 384, 9), (37, 21, 9), (33, 65, 4), (5, 3, 1)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pack_and_unpack_conv_weight(Co, Ci, taps, dtype):
    lib, check, ptr, stream = _lib()
    from htrvt_amd.ops import cpad, dt
    g = torch.Generator().manual_seed(Co + Ci + taps)
    w = torch.randint(-64, 64, (Co, Ci, taps), generator=g).float().cuda()
    cpi, cpo = cpad(Ci, dtype), cpad(Co, dtype)
    fwd = torch.zeros(Co, taps, cpi, dtype=dtype, device="cuda")
    dgr = torch.zeros(Ci, taps, cpo, dtype=dtype, device="cuda")
    check(lib.htrvt_pack_conv_weight(ptr(w), ptr(fwd), ptr(dgr), Co, Ci, taps, cpi, cpo, dt(dtype), stream()), "pack")
    assert torch.equal(fwd[:, :, :Ci].float(), w.permute(0, 2, 1))
    assert torch.equal(dgr[:, :, :Co].float(), w.permute(1, 2, 0))
    assert (fwd[:, :, Ci:] == 0).all() and (dgr[:, :, Co:] == 0).all()
    packed = torch.randint(-64, 64, (taps, cpi, Co), generator=g).float().cuda()
    grad = w.clone()
    check(lib.htrvt_unpack_conv_wgrad(ptr(packed), ptr(grad), Co, Ci, taps, cpi, stream()), "unpack")
    assert torch.equal(grad, w + packed[:, :Ci, :].permute(2, 1, 0))


def test_relayout_table_runs_every_job_of_a_launch():
    """csrc/relayout.hip: conv packs (one of them into the tap slots of a joint buffer), Linear cast + transpose (one with
    the head's padded leading dimension) and weight-gradient unpacks as ONE table-driven launch each, against torch."""
    lib, check, ptr, stream = _lib()
    import ctypes as C
    from htrvt_amd._lib import RelayoutJob, RELAYOUT_PACK_CONV, RELAYOUT_CAST_TRANSPOSE, RELAYOUT_UNPACK_WGRAD
    from htrvt_amd.engine import _job
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    ints = lambda *shape: torch.randint(-64, 64, shape, generator=g).float().cuda()

    calls = [0]

    def run(jobs, dti):
        arr = (RelayoutJob * len(jobs))(*jobs)
        total = lib.htrvt_relayout_plan(arr, len(jobs))
        assert total > 0, lib.htrvt_last_error()
        if calls[0] % 2 == 0:       # table in device memory ...
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
            check(lib.htrvt_relayout(ptr(dev), len(jobs), total, dti, stream()), "relayout")
        else:                       # ... or copied into the kernel-argument segment (what the engine does)
            check(lib.htrvt_relayout_host(arr, len(jobs), total, dti, stream()), "relayout_host")
        calls[0] += 1
        torch.cuda.synchronize()

    convs = [(192, 64, 9), (40, 24, 9), (384, 192, 1), (37, 21, 9), (16, 8, 4)]
    jobs, want = [], []
    for Co, Ci, taps in convs:
        w = ints(Co, Ci, taps)
        cpi, cpo = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
        fwd = torch.zeros(Co, taps, cpi, dtype=bf, device="cuda")
        dgr = torch.zeros(Ci, taps, cpo, dtype=bf, device="cuda")
        jobs.append(_job(RELAYOUT_PACK_CONV, w, fwd, dgr, Co, Ci, taps, cpi, cpo, taps, 0))
        want.append((w, fwd, dgr, Co, Ci))
    # a 3x3 and a 1x1 weight sharing one [Ci][10][Co] dgrad buffer (HtrvtGemmDesc.A2)
    w3, w1 = ints(48, 24, 9), ints(48, 24, 1)
    joint = torch.zeros(24, 10, 48, dtype=bf, device="cuda")
    jobs.append(_job(RELAYOUT_PACK_CONV, w3, None, joint, 48, 24, 9, 24, 48, 10, 0))
    jobs.append(_job(RELAYOUT_PACK_CONV, w1, None, joint, 48, 24, 1, 24, 48, 10, 9))
    lins = []
    for rows, cols, ld in [(2304, 768, 2304), (80, 768, 80), (77, 100, 80)]:
        w = ints(rows, cols)
        d0 = torch.zeros(rows, cols, dtype=bf, device="cuda")
        d1 = torch.full((cols, ld), 7.0, dtype=bf, device="cuda")
        jobs.append(_job(RELAYOUT_CAST_TRANSPOSE, w, d0, d1, rows, cols, 0, ld))
        lins.append((w, d0, d1, rows))
    run(jobs, 1)
    for _, fwd_, dgr_, _, _ in want:      # second pass through the other launch form, from cleared destinations
        fwd_.zero_(), dgr_.zero_()
    joint.zero_()
    run(jobs, 1)
    for w, fwd, dgr, Co, Ci in want:
        assert torch.equal(fwd[:, :, :Ci].float(), w.permute(0, 2, 1)) and torch.equal(dgr[:, :, :Co].float(), w.permute(1, 2, 0))
        assert (fwd[:, :, Ci:] == 0).all() and (dgr[:, :, Co:] == 0).all()
    assert torch.equal(joint[:, :9].float(), w3.permute(1, 2, 0)) and torch.equal(joint[:, 9].float(), w1[:, :, 0].t())
    for w, d0, d1, rows in lins:
        assert torch.equal(d0.float(), w) and torch.equal(d1[:, :rows].float(), w.t()) and (d1[:, rows:] == 0).all()
    # unpack: grad += packed^T for all convs in one launch
    jobs, want = [], []
    for Co, Ci, taps in convs:
        cpi = (Ci + 7) // 8 * 8
        packed, grad = ints(taps, cpi, Co), ints(Co, Ci, taps)
        want.append((grad.clone() + packed[:, :Ci, :].permute(2, 1, 0), grad, packed))
        jobs.append(_job(RELAYOUT_UNPACK_WGRAD, packed, grad, None, Co, Ci, taps, cpi))
    run(jobs, 1)
    for w_, got, _ in want:
        assert torch.equal(got, w_)
    # a malformed job is refused by the plan, not launched
    bad = (RelayoutJob * 1)(_job(RELAYOUT_PACK_CONV, w3, None, joint, 48, 24, 9, 24, 48, 9, 1))     # tap slots 1..9 of 9
    assert lib.htrvt_relayout_plan(bad, 1) < 0


@pytest.mark.parametrize("rows,cols,ld", [(32768, 768, 768), (1000, 80, 88), (37, 3072, 3072), (4096, 2304, 2304)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_colsum(rows, cols, ld, dtype):
    lib, check, ptr, stream = _lib()
    from htrvt_amd.ops import colsum, dt
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randint(-3, 4, (rows, ld), generator=g).to(dtype).cuda()
    out = torch.full((cols,), 5.0, device="cuda")
    colsum(x, rows, cols, ld, out, dti=dt(dtype))
    assert torch.equal(out, 5.0 + x[:, :cols].float().sum(0))
    # row filter: only rows r with keep[r % N] == 0 contribute (gradient of the mask token)
    N = 16
    keep = (torch.arange(N) % 3 != 0).float().cuda()
    out2 = torch.zeros(cols, device="cuda")
    colsum(x, rows, cols, ld, out2, dti=dt(dtype), keep=keep, keep_mod=N)
    sel = keep[torch.arange(rows, device="cuda") % N] == 0
    assert torch.equal(out2, x[sel][:, :cols].float().sum(0))


@pytest.mark.parametrize("H,W,D", [(2, 256, 768), (2, 37, 64), (4, 48, 64), (5, 16, 32)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pool_tokens_and_its_backward_match_max_pool2d(H, W, D, dtype):
    """final max_pool2d(3, stride (2,1), pad 1) + span mask + pos-embed (HTR_VT.py:226-236) and its backward, through
    the C ABI, against torch's max_pool2d autograd on the CPU; H = 2 takes the one-pooled-row kernel, H > 2 the general
    one."""
    lib, check, ptr, stream = _lib()
    from htrvt_amd.ops import dt
    B = 3
    g = torch.Generator().manual_seed(H * 1000 + W)
    Ho = (H - 1) // 2 + 1
    N = Ho * W
    # every 3x3 window sees the nine cells (h % 3, w % 3) once: a per-(image, channel) random permutation of 0..8 laid out
    # periodically has a unique maximum in every window (clipped ones too) and is exact in bfloat16
    perm = torch.stack([torch.randperm(9, generator=g) for _ in range(B * D)]).view(B, D, 9).float()
    cell = (torch.arange(H).view(H, 1) % 3) * 3 + (torch.arange(W).view(1, W) % 3)
    x = perm[:, :, cell.view(-1)].view(B, D, H, W)
    keep = (torch.rand(N, generator=g) > 0.3).float()
    dtok = torch.randint(-8, 9, (B, N, D), generator=g).float()
    xr = x.clone().requires_grad_(True)
    pooled = torch.nn.functional.max_pool2d(xr, kernel_size=3, stride=(2, 1), padding=1)   # [B, D, Ho, W]
    tok_ref = pooled.permute(0, 2, 3, 1).reshape(B, N, D)
    (tok_ref * keep.view(1, N, 1) * dtok).sum().backward()
    dx_ref = xr.grad.permute(0, 2, 3, 1).contiguous()       # NHWC, as the kernels see it

    x_d = x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    keep_d, dtok_d = keep.cuda(), dtok.to(dtype).cuda()     # named: a temporary would be freed (and reused) before the launch
    mask_token = torch.full((D,), -7.0).cuda()
    pos = torch.zeros(N, D).cuda()
    tok = torch.empty(B, N, D, dtype=dtype, device="cuda")
    check(lib.htrvt_pool_tokens(ptr(x_d), ptr(keep_d), ptr(mask_token), ptr(pos), ptr(tok), B, H, N, D, dt(dtype), stream()),
          "pool_tokens")
    want = torch.where(keep.view(1, N, 1) != 0, tok_ref.detach(), torch.tensor(-7.0))
    assert torch.equal(tok.float().cpu(), want)
    dx = torch.empty(B, H, W, D, dtype=dtype, device="cuda")
    check(lib.htrvt_pool_tokens_bwd(ptr(dtok_d), ptr(x_d), ptr(keep_d), ptr(dx), B, H, N, D, dt(dtype), stream()),
          "pool_tokens_bwd")
    assert torch.equal(dx.float().cpu(), dx_ref)


@pytest.mark.parametrize("rows,cols", [(4096, 768), (1000, 200), (37, 1728), (3, 9)])
def test_split_bf16_forms(rows, cols):
    """htrvt_split_bf16 (csrc/split.hip): hi = bf16(x), lo = bf16(x - hi), the concatenated forms in both orders, the
    transposed form, the float32 copy of the parts, the planes -- exact against torch's own bfloat16 rounding; and
    hi + lo within 2^-16 |x| of x (what the split-bf16 parity path relies on)"""
    from htrvt_amd._lib import check, lib
    from htrvt_amd.ops import ptr, stream
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * torch.logspace(-3, 3, cols)).cuda()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    assert ((hi.float() + lo.float()) - x).abs().max() <= 2.0 ** -16 * x.abs().max()
    BF = torch.bfloat16
    for order, parts in ((0, (hi, lo, hi)), (1, (hi, hi, lo))):
        want = torch.cat(parts, dim=1)
        cat = torch.empty(rows, 3 * cols, dtype=BF, device="cuda")
        check(lib.htrvt_split_bf16(ptr(x), rows, cols, cols, ptr(cat), order, 0, 0, None, None, stream()))
        assert torch.equal(cat, want)
        catf = torch.empty(rows, 3 * cols, dtype=torch.float32, device="cuda")
        check(lib.htrvt_split_bf16(ptr(x), rows, cols, cols, ptr(catf), order, 1, 0, None, None, stream()))
        assert torch.equal(catf, want.float())
        catt = torch.empty(cols, 3 * rows, dtype=BF, device="cuda")
        check(lib.htrvt_split_bf16(ptr(x), rows, cols, cols, ptr(catt), order, 0, 1, None, None, stream()))
        assert torch.equal(catt, torch.cat([p.t() for p in parts], dim=1))
    if cols % 8 == 0:
        h2, l2 = torch.empty(rows, cols, dtype=BF, device="cuda"), torch.empty(rows, cols, dtype=BF, device="cuda")
        check(lib.htrvt_split_bf16(ptr(x), rows, cols, cols, None, 0, 0, 0, ptr(h2), ptr(l2), stream()))
        assert torch.equal(h2, hi) and torch.equal(l2, lo)


def test_elementwise_f32_steps():
    """htrvt_elementwise_f32: the float32 GELU / GELU' / residual steps beside the split-bf16 Linear products, against
    torch's float64 erf GELU and its autograd derivative"""
    from htrvt_amd._lib import check, lib
    from htrvt_amd.ops import ptr, stream
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(1000, 768, generator=g) * 2).cuda()
    b = (torch.randn(1000, 768, generator=g) * 2).cuda()
    out = torch.empty_like(a)
    check(lib.htrvt_elementwise_f32(ptr(a), None, ptr(out), a.numel(), 0, stream()))
    assert (out.double() - torch.nn.functional.gelu(a.double())).abs().max() < 2e-6
    bd = b.double().requires_grad_(True)
    torch.nn.functional.gelu(bd).sum().backward()
    check(lib.htrvt_elementwise_f32(ptr(a), ptr(b), ptr(out), a.numel(), 1, stream()))
    assert (out.double() - a.double() * bd.grad).abs().max() < 1e-5
    check(lib.htrvt_elementwise_f32(ptr(a), ptr(b), ptr(out), a.numel(), 2, stream()))
    assert torch.equal(out, a + b)
    check(lib.htrvt_elementwise_f32(ptr(out), ptr(b), ptr(out), a.numel(), 2, stream()))      # in place
    assert torch.equal(out, (a + b) + b)


@pytest.mark.parametrize("sh,sw,Hi,Wi", [(2, 2, 8, 64), (2, 1, 16, 32), (2, 2, 7, 33)])
def test_class_scatter_f32(sh, sw, Hi, Wi):
    """htrvt_class_scatter_f32: dense per-parity-class matrices -> the NHWC gradient (+ residual), odd extents included"""
    from htrvt_amd._lib import check, lib
    from htrvt_amd.ops import ptr, stream
    B, C = 3, 24
    g = torch.Generator().manual_seed(sh * 10 + sw)
    full = torch.randn(B, Hi, Wi, C, generator=g).cuda()
    res = torch.randn(B, Hi, Wi, C, generator=g).cuda()
    parts = {(a, b): full[:, a::sh, b::sw, :].contiguous() for a in range(sh) for b in range(sw)}
    dx = torch.full_like(full, float("nan"))
    for r in (None, res):
        check(lib.htrvt_class_scatter_f32(ptr(parts[(0, 0)]), ptr(parts.get((0, 1))), ptr(parts.get((1, 0))), ptr(parts.get((1, 1))),
                                          ptr(r), ptr(dx), B, Hi, Wi, C, sh, sw, stream()))
        assert torch.equal(dx, full if r is None else full + res)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2, 8, 256, 192), (1, 3, 37, 384)])
def test_bn_apply_mask_is_the_sign_of_the_output(mode, shape):
    """htrvt_bn_apply_mask = htrvt_bn_apply + the 1-bit ReLU mask of the output (one byte per 8 channels, bit j = element
    8 i + j > 0): the output must equal the mask-free pass bit for bit and the mask must be the sign of the STORED output"""
    import numpy as np
    import htrvt_amd  # noqa: F401
    from htrvt_amd._lib import check, lib
    from htrvt_amd.ops import dt, ptr, stream
    torch.manual_seed(5)
    bf = dt(torch.bfloat16)
    C = shape[-1]
    x = torch.randn(*shape, device="cuda").bfloat16()
    res = torch.randn(*shape, device="cuda").bfloat16() if mode else None
    sc, sf = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    rsc, rsf = (torch.randn(C, device="cuda"), torch.randn(C, device="cuda")) if mode == 2 else (None, None)
    npix = x.numel() // C
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    mask = torch.zeros(x.numel() // 8, dtype=torch.uint8, device="cuda")
    check(lib.htrvt_bn_apply(ptr(x), ptr(sc), ptr(sf), ptr(res), ptr(rsc), ptr(rsf), ptr(y0), npix, C, 1, bf, stream()), "bn_apply")
    check(lib.htrvt_bn_apply_mask(ptr(x), ptr(sc), ptr(sf), ptr(res), ptr(rsc), ptr(rsf), ptr(y1), ptr(mask), npix, C, 1, bf, stream()),
          "bn_apply_mask")
    assert torch.equal(y0, y1)
    want = np.packbits((y0.float() > 0).cpu().numpy().reshape(-1), bitorder="little")
    assert np.array_equal(mask.cpu().numpy(), want)
    frac = float((y0.float() > 0).float().mean())
    assert 0.2 < frac < 0.8
