"""Attention flavours of the reference's variant models on the hot path's kernels, against literal
float64 restatements of the reference modules:
  * model_window/model/HTR_VT.py:33-62 (Attention.forward with the relative-position bias) and :113-154
    (Block._attend: roll, 1-D window partition, per-window attention, reverse) -- restated below step by step, while
    the implementation under test builds ONE dense [heads, N, N] bias and runs a single fused launch;
  * model_sgm_2/model/sgm_head.py:118-127 (SGMHead._cross_attend)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _window_reference(qkv, table, B, N, h, hd, P, ws, shift):
    """float64, the reference's own sequence of operations on the per-token q/k/v (the qkv Linear commutes with the roll
    and the partition, which only move whole tokens)"""
    D = h * hd
    x = qkv.double().reshape(B, N, 3 * D)
    table = table.double()

    def attn(xw):                                      # Attention.forward on [B', n, 3D] (n = N or ws)
        Bp, n, _ = xw.shape
        q, k, v = xw.reshape(Bp, n, 3, h, hd).permute(2, 0, 3, 1, 4).unbind(0)
        a = (q @ k.transpose(-2, -1)) * hd ** -0.5
        coords = torch.arange(P)
        idx = (coords[None, :] - coords[:, None]) + P - 1
        a = a + table[idx[:n, :n]].permute(2, 0, 1).unsqueeze(0)
        return (a.softmax(-1) @ v).transpose(1, 2).reshape(Bp, n, D)

    if ws <= 0:
        return attn(x).reshape(B * N, D)
    xs = torch.roll(x, shifts=(-shift,), dims=1) if shift > 0 else x
    xw = xs.reshape(B * (N // ws), ws, 3 * D)
    y = attn(xw).reshape(B, N, D)
    if shift > 0:
        y = torch.roll(y, shifts=(shift,), dims=1)
    return y.reshape(B * N, D)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("ws,shift", [(0, 0), (16, 0), (16, 8)])
def test_window_relative_bias_attention_forward_backward(dtype, ws, shift):
    from htrvt_amd import variants as V
    B, N, h, hd, P = 3, 128, 4, 64, 128
    D = h * hd
    g = torch.Generator().manual_seed(17 + ws + shift)
    qkv = (torch.randn(B * N, 3 * D, generator=g)).to(dtype)
    table = (torch.randn(2 * P - 1, h, generator=g) * 0.5)
    dout = torch.randn(B * N, D, generator=g).to(dtype)

    qr = qkv.double().clone().requires_grad_(True)
    tr = table.double().clone().requires_grad_(True)
    ref = _window_reference(qr, tr, B, N, h, hd, P, ws, shift)
    ref.backward(dout.double())

    qd = qkv.cuda().requires_grad_(True)
    td = table.cuda().requires_grad_(True)
    bias = V.relative_position_bias(td, N, P, ws, shift)          # HIP kernel (csrc/variants.hip), differentiable in the table
    out = V.biased_self_attention(qd, bias, B, N, h)
    out.backward(dout.cuda())
    f32 = dtype == torch.float32
    err = (out.detach().double().cpu() - ref.detach()).abs().max().item()
    assert err < (2e-5 if f32 else 2.5e-2), err
    for name, got, want in (("dqkv", qd.grad, qr.grad), ("dtable", td.grad, tr.grad)):
        got, want = got.double().cpu(), want
        e = (got - want).abs().max().item() / want.abs().max().item()
        cos = float((got.flatten() @ want.flatten()) / (got.norm() * want.norm()))
        print(f"{dtype} ws={ws} shift={shift} {name}: rel-to-max {e:.3e} cosine {cos:.6f}")
        assert e < (1e-4 if f32 else 3e-2) and cos > (0.999999 if f32 else 0.9995), (name, e, cos)
    # entries of the table no (query, key) pair uses get exactly zero gradient
    idx, inside = V.relative_position_index(N, P, ws, shift)
    used = torch.zeros(2 * P - 1, dtype=torch.bool)
    used[idx[inside]] = True
    assert float(td.grad.cpu()[~used].abs().max() if (~used).any() else 0.0) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("L,N,D", [(37, 128, 768), (64, 256, 256), (5, 64, 64)])
def test_sgm_cross_attention_forward_backward(dtype, L, N, D):
    from htrvt_amd import variants as V
    B = 3
    g = torch.Generator().manual_seed(L + N)
    Q = torch.randn(B, L, D, generator=g).to(dtype)
    F = torch.randn(B, N, D, generator=g).to(dtype)
    dout = torch.randn(B, L, D, generator=g).to(dtype)
    qr, fr = Q.double().clone().requires_grad_(True), F.double().clone().requires_grad_(True)
    attn = torch.einsum("bld,bnd->bln", qr, fr) / (D ** 0.5)          # sgm_head.py:122-126
    ref = torch.einsum("bln,bnd->bld", attn.softmax(-1), fr)
    ref.backward(dout.double())
    qd, fd = Q.cuda().requires_grad_(True), F.cuda().requires_grad_(True)
    out = V.cross_attention(qd, fd)
    out.backward(dout.cuda())
    f32 = dtype == torch.float32
    assert (out.detach().double().cpu() - ref.detach()).abs().max() < (2e-5 if f32 else 3e-2)
    for name, got, want in (("dQ", qd.grad, qr.grad), ("dKV", fd.grad, fr.grad)):
        got = got.double().cpu()
        e = (got - want).abs().max().item() / want.abs().max().item()
        cos = float((got.flatten() @ want.flatten()) / (got.norm() * want.norm()))
        print(f"{dtype} L={L} N={N} D={D} {name}: rel-to-max {e:.3e} cosine {cos:.6f}")
        assert e < (1e-4 if f32 else 3e-2) and cos > (0.999999 if f32 else 0.999), (name, e, cos)


# --------------------------------------------------------------------------------------------------------------------
# pinned by the reference: tools/make_goldens_variants.py ran model_window's Block._attend / Attention.forward and
# model_sgm_2's SGMHead._cross_attend (float64, CPU) on the seeded inputs of tests/variant_cases.py
# --------------------------------------------------------------------------------------------------------------------
import os          # noqa: E402

import numpy as np     # noqa: E402

import variant_cases as VC    # noqa: E402


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", VC.WINDOW_CASES, ids=lambda c: c[0])
def test_window_attention_against_reference_fixture(golden_dir, dtype, case):
    """x -> qkv Linear -> [relative bias (+ windows, shift, padding) attention: the kernels] -> proj, forward and the
    gradients of x, the bias table and the qkv bias, against the reference module's own output.  The two Linears around
    the kernels are torch glue here (the model's Linears are htrvt_gemm; tests/test_model_gpu.py covers those)."""
    from htrvt_amd import variants as V
    tag, B, N, dim, heads, P, ws, shift = case
    g = np.load(os.path.join(golden_dir, "variants.npz"))
    inp = {k: torch.from_numpy(v).float().cuda() for k, v in VC.window_inputs(case).items()}
    x = inp["x"].clone().requires_grad_(True)
    table = inp["table"].clone().requires_grad_(True)
    qb = inp["qkv_b"].clone().requires_grad_(True)
    hd = dim // heads
    qkv = (x.reshape(B * N, dim) @ inp["qkv_w"].t() + qb).to(dtype)
    ld = V.padded_len(N, dtype, hd)
    bias = V.relative_position_bias(table, N, P, ws, shift, ld=ld)
    core = V.biased_self_attention(qkv, bias, B, N, heads)
    y = core.float() @ inp["proj_w"].t() + inp["proj_b"]
    y.backward(inp["gout"].reshape(B * N, dim))
    f32 = dtype == torch.float32
    want = torch.from_numpy(g[f"win.{tag}.y"]).reshape(B * N, dim)
    err = (y.detach().cpu() - want).abs().max().item() / want.abs().max().item()
    print(f"{tag} {dtype}: ld {ld}, y rel-to-max {err:.3e}")
    assert err < (2e-5 if f32 else 2e-2), err
    for name, got in (("dx", x.grad.reshape(B, N, dim)), ("dtable", table.grad), ("dqkv_b", qb.grad)):
        w_ = torch.from_numpy(g[f"win.{tag}.{name}"])
        got = got.cpu()
        e = (got - w_).abs().max().item() / w_.abs().max().item()
        cos = float((got.flatten().double() @ w_.flatten().double()) / (got.double().norm() * w_.double().norm()))
        print(f"   {name}: rel-to-max {e:.3e} cosine {cos:.6f}")
        assert e < (1e-4 if f32 else 4e-2) and cos > (0.999999 if f32 else 0.999), (name, e, cos)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", VC.SGM_CASES, ids=lambda c: c[0])
def test_sgm_cross_attention_against_reference_fixture(golden_dir, dtype, case):
    """kv LayerNorm (htrvt_layernorm_fwd/bwd are the model's; torch glue here) -> cross-attention kernels, vs SGMHead._cross_attend"""
    from htrvt_amd import variants as V
    tag, B, L, N, D = case
    g = np.load(os.path.join(golden_dir, "variants.npz"))
    inp = {k: torch.from_numpy(v).float().cuda() for k, v in VC.sgm_inputs(case).items()}
    Q = inp["Q"].clone().requires_grad_(True)
    Fv = inp["F"].clone().requires_grad_(True)
    K = torch.nn.functional.layer_norm(Fv, (D,), inp["ln_w"], inp["ln_b"], 1e-5)
    y = V.cross_attention(Q.to(dtype), K.to(dtype)).float()
    y.backward(inp["gout"])
    f32 = dtype == torch.float32
    want = torch.from_numpy(g[f"sgm.{tag}.y"])
    assert (y.detach().cpu() - want).abs().max().item() < (3e-5 if f32 else 3e-2) * want.abs().max().item()
    for name, got in (("dQ", Q.grad), ("dF", Fv.grad)):
        w_ = torch.from_numpy(g[f"sgm.{tag}.{name}"])
        got = got.cpu()
        e = (got - w_).abs().max().item() / w_.abs().max().item()
        cos = float((got.flatten().double() @ w_.flatten().double()) / (got.double().norm() * w_.double().norm()))
        print(f"{tag} {dtype} {name}: rel-to-max {e:.3e} cosine {cos:.6f}")
        assert e < (2e-4 if f32 else 4e-2) and cos > (0.99999 if f32 else 0.999), (name, e, cos)
