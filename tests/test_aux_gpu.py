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


@pytest.mark.parametrize("Co,Ci,taps", [(192, 192, 9), (384, 192, 1), (40, 24, 9), (768, 384, 9)])
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
