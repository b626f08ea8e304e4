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


def _dense(table, N, P, ws, shift):
    """the dense bias the kernels build (csrc/variants.hip), from the host-side index bookkeeping"""
    from htrvt_amd import variants as V
    idx, inside = V.relative_position_index(N, P, ws, shift)
    b = table[idx].permute(2, 0, 1)
    return torch.where(inside[None], b, torch.full_like(b, V.MASKED))


@pytest.mark.parametrize("ws,shift", [(0, 0), (16, 0), (16, 8), (8, 3)])
def test_dense_bias_equals_window_procedure(ws, shift):
    B, N, h, hd, P = 2, 64, 3, 8, 80
    g = torch.Generator().manual_seed(ws * 10 + shift)
    qkv = torch.randn(B * N, 3 * h * hd, generator=g, dtype=torch.float64)
    table = torch.randn(2 * P - 1, h, generator=g, dtype=torch.float64)
    want = _literal(qkv, table, B, N, h, hd, P, ws, shift)
    bias = _dense(table, N, P, ws, shift)
    assert bias.shape == (h, N, N)
    q, k, v = qkv.reshape(B, N, 3, h, hd).permute(2, 0, 3, 1, 4).unbind(0)
    got = (((q @ k.transpose(-2, -1)) * hd ** -0.5 + bias[None]).softmax(-1) @ v).transpose(1, 2).reshape(B, N, h * hd)
    assert (got - want).abs().max() < 1e-9


@pytest.mark.parametrize("case", __import__("variant_cases").WINDOW_CASES, ids=lambda c: c[0])
def test_dense_bias_reproduces_the_reference_block(golden_dir, case):
    """pinned by the reference itself (tools/make_goldens_variants.py ran model_window's Block._attend): a plain float64
    softmax over the ONE dense bias gives the reference's output and gradients, including N not a multiple of the window
    (zero padding + key_padding_mask in the reference)"""
    import os
    import numpy as np
    import variant_cases as VC
    from htrvt_amd import variants as V
    tag, B, N, dim, heads, P, ws, shift = case
    g = np.load(os.path.join(golden_dir, "variants.npz"))
    inp = {k: torch.from_numpy(v) for k, v in VC.window_inputs(case).items()}
    x = inp["x"].clone().requires_grad_(True)
    table = inp["table"].clone().requires_grad_(True)
    hd = dim // heads
    qkv = x @ inp["qkv_w"].t() + inp["qkv_b"]
    q, k, v = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
    idx, inside = V.relative_position_index(N, P, ws, shift)
    bias = torch.where(inside[None], table[idx].permute(2, 0, 1), torch.full((heads, N, N), V.MASKED, dtype=torch.float64))
    core = (((q @ k.transpose(-2, -1)) * hd ** -0.5 + bias[None]).softmax(-1) @ v).transpose(1, 2).reshape(B, N, dim)
    y = core @ inp["proj_w"].t() + inp["proj_b"]
    y.backward(inp["gout"])
    assert (y.detach() - torch.from_numpy(g[f"win.{tag}.y"]).double()).abs().max() < 2e-6
    assert (x.grad - torch.from_numpy(g[f"win.{tag}.dx"]).double()).abs().max() < 2e-5
    assert (table.grad - torch.from_numpy(g[f"win.{tag}.dtable"]).double()).abs().max() < 2e-5
