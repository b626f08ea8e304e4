"""Index bookkeeping of htrvt_amd.variants on the CPU (no kernel runs): the ONE dense bias [heads, N, N] that stands in
for model_window's roll / window-partition / per-window attention / reverse (model_window/model/HTR_VT.py:113-154) and its
relative-position table lookup (:23-31,45-46) must reproduce that procedure exactly when used in a plain softmax."""
import pytest
import torch


def _literal(qkv, table, B, N, h, hd, P, ws, shift):
    D = h * hd
    x = qkv.reshape(B, N, 3 * D)

    def attn(xw):
        Bp, n, _ = xw.shape
        q, k, v = xw.reshape(Bp, n, 3, h, hd).permute(2, 0, 3, 1, 4).unbind(0)
        a = (q @ k.transpose(-2, -1)) * hd ** -0.5
        coords = torch.arange(P)
        idx = (coords[None, :] - coords[:, None]) + P - 1
        a = a + table[idx[:n, :n]].permute(2, 0, 1).unsqueeze(0)
        return (a.softmax(-1) @ v).transpose(1, 2).reshape(Bp, n, D)

    if ws <= 0:
        return attn(x)
    xs = torch.roll(x, shifts=(-shift,), dims=1) if shift > 0 else x
    y = attn(xs.reshape(B * (N // ws), ws, 3 * D)).reshape(B, N, D)
    return torch.roll(y, shifts=(shift,), dims=1) if shift > 0 else y


@pytest.mark.parametrize("ws,shift", [(0, 0), (16, 0), (16, 8), (8, 3)])
def test_dense_bias_equals_window_procedure(ws, shift):
    from htrvt_amd import variants as V
    B, N, h, hd, P = 2, 64, 3, 8, 80
    g = torch.Generator().manual_seed(ws * 10 + shift)
    qkv = torch.randn(B * N, 3 * h * hd, generator=g, dtype=torch.float64)
    table = torch.randn(2 * P - 1, h, generator=g, dtype=torch.float64)
    want = _literal(qkv, table, B, N, h, hd, P, ws, shift)
    bias = V.relative_position_bias(table, N, P, ws, shift).double()
    assert bias.shape == (h, N, N)
    q, k, v = qkv.reshape(B, N, 3, h, hd).permute(2, 0, 3, 1, 4).unbind(0)
    got = (((q @ k.transpose(-2, -1)) * hd ** -0.5 + bias[None]).softmax(-1) @ v).transpose(1, 2).reshape(B, N, h * hd)
    assert (got - want).abs().max() < 1e-6        # the dense bias is float32 (what the kernels take)
    # gradient gather: d(table) through the dense bias == autograd through the table lookup
    t = table.clone().requires_grad_(True)
    (V.relative_position_bias(t, N, P, ws, shift).double() * torch.where(bias > -1e29, torch.ones_like(bias), torch.zeros_like(bias))
     ).sum().backward()
    dense = torch.where(bias > -1e29, torch.ones_like(bias), torch.zeros_like(bias)).float()
    gathered = V.relative_position_bias_grad(dense, table.shape, N, P, ws, shift)
    assert torch.allclose(gathered.double(), t.grad, atol=1e-5)


def test_window_size_must_divide():
    from htrvt_amd import variants as V
    with pytest.raises(ValueError):
        V.relative_position_index(100, 128, window_size=16)
