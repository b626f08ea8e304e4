"""Fused attention kernels (csrc/attention.hip) through the C ABI against a float64 restatement of
reference model_v1/model/HTR_VT.py:27-36 (softmax(q k^T * scale) v) and torch autograd of it.

Structured cases first (one-hot / integer operands: any error in a fragment layout, in the accumulator-as-operand k
order or in the LDS image shows as an O(1) difference in a known place), then random data at the shapes the models
use: (N, hd) = (256, 128) headline, (128, 128) d768 @ W=512, (512, 128) long line, (256 / 128, 64) d512 / d256,
(128, 32) tiny model."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(2, 256, 6, 128), (3, 128, 6, 128), (1, 512, 6, 128), (2, 256, 8, 64), (2, 128, 4, 64), (4, 128, 2, 32),
          # sequence lengths that are not multiples of the 128-query / 64-key tiles (W = 576 -> N = 144, ...): partial
          # last tiles, masked padding keys / queries (reference: HTR_VT.py:27-39 has no length restriction)
          (2, 144, 6, 128), (3, 200, 4, 64), (2, 72, 2, 32), (1, 328, 6, 128), (2, 40, 2, 64)]


def _lib():
    import htrvt_amd  # noqa: F401
    from htrvt_amd._lib import check, lib
    from htrvt_amd.ops import ptr, stream
    return lib, check, ptr, stream


def _ref(qkv, B, N, h, hd, dout=None):
    """float64 attention over the [B,N,3,h,hd] layout (operands as given, i.e. already bf16-rounded)"""
    x = qkv.double().reshape(B, N, 3, h, hd).permute(2, 0, 3, 1, 4).clone().requires_grad_(dout is not None)
    q, k, v = x[0], x[1], x[2]
    s = (q @ k.transpose(-1, -2)) * hd ** -0.5
    p = s.softmax(-1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * N, h * hd)
    lse2 = torch.logsumexp(s, -1) * math.log2(math.e)          # [B,h,N]
    if dout is None:
        return o, lse2, None
    o.backward(dout.double())
    dqkv = x.grad.permute(1, 3, 0, 2, 4).reshape(B * N, 3 * h * hd)
    return o.detach(), lse2.detach(), dqkv


def _fwd(qkv, B, N, h, hd, want_lse=True, bias=None):
    lib, check, ptr, stream = _lib()
    out = torch.full((B * N, h * hd), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.full((B * h, N), float("nan"), dtype=torch.float32, device="cuda") if want_lse else None
    check(lib.htrvt_attn_fwd(ptr(qkv), ptr(bias), ptr(out), ptr(lse), B, N, h, hd, hd ** -0.5, 1, stream()), "attn_fwd")
    return out, lse


def _bwd(qkv, out, dout, lse, B, N, h, hd, bias=None, dbias=None):
    lib, check, ptr, stream = _lib()
    dqkv = torch.full((B * N, 3 * h * hd), float("nan"), dtype=torch.bfloat16, device="cuda")
    delta = torch.empty(B * h, N, dtype=torch.float32, device="cuda")
    check(lib.htrvt_attn_bwd(ptr(qkv), ptr(bias), ptr(out), ptr(dout), ptr(lse), ptr(delta), ptr(dqkv), ptr(dbias), B, N, h, hd, hd ** -0.5, 1,
                             stream()), "attn_bwd")
    return dqkv, delta


def test_supported_predicate():
    lib, _, _, _ = _lib()
    assert lib.htrvt_attn_supported(256, 128, 1) and lib.htrvt_attn_supported(128, 32, 1) and lib.htrvt_attn_supported(512, 64, 1)
    assert not lib.htrvt_attn_supported(256, 128, 0)      # float32 parity path: batched GEMMs + row softmax
    assert lib.htrvt_attn_supported(64, 128, 1) and lib.htrvt_attn_supported(192, 64, 1) and lib.htrvt_attn_supported(144, 128, 1)
    assert not lib.htrvt_attn_supported(256, 96, 1) and not lib.htrvt_attn_supported(16, 64, 1)


@pytest.mark.parametrize("B,N,h,hd", SHAPES)
def test_forward_one_hot_attention_copies_the_selected_value_row(B, N, h, hd):
    """q[i] and k[perm[i]] share a large one-hot direction, so softmax row i is (to 1e-9) one-hot at key perm[i] and the
    output must be EXACTLY v[perm[i]] (integers): catches any row / key / head / d permutation."""
    g = torch.Generator().manual_seed(N + hd)
    qkv = torch.zeros(B, N, 3, h, hd)
    perm = torch.stack([torch.stack([torch.randperm(N, generator=g) for _ in range(h)]) for _ in range(B)])   # [B,h,N]
    # codes: query i and key perm[i] get the same +-1 code word scaled up; distinct codes are far apart
    code = (torch.randint(0, 2, (N, hd), generator=g) * 2 - 1).float()
    for b in range(B):
        for hh in range(h):
            qkv[b, :, 0, hh, :] = code * 8.0
            qkv[b, perm[b, hh], 1, hh, :] = code * 8.0
    qkv[:, :, 2] = torch.randint(-8, 9, (B, N, h, hd), generator=g).float()
    d = qkv.to(torch.bfloat16).cuda().reshape(B * N, 3 * h * hd)
    out, lse = _fwd(d, B, N, h, hd)
    v = qkv[:, :, 2]                                                       # [B,N,h,hd]
    want = torch.stack([torch.stack([v[b, perm[b, hh], hh, :] for hh in range(h)], 1) for b in range(B)])   # [B,N,h,hd]
    # the matching key scores 64 hd * scale, any other code word at most ~half of that: the leak is < 1e-9
    ref_o, ref_lse, _ = _ref(d.cpu(), B, N, h, hd)
    assert (ref_o - want.reshape(B * N, h * hd).double()).abs().max() < 1e-6     # the construction itself
    assert torch.equal(out.float().cpu(), want.reshape(B * N, h * hd))
    assert (lse.double().cpu().reshape(B, h, N) - ref_lse).abs().max() < 1e-3


@pytest.mark.parametrize("B,N,h,hd", SHAPES)
def test_forward_random_matches_float64(B, N, h, hd):
    g = torch.Generator().manual_seed(7 * N + hd)
    qkv = (torch.randn(B * N, 3 * h * hd, generator=g) * 1.5).to(torch.bfloat16)
    out, lse = _fwd(qkv.cuda(), B, N, h, hd)
    ref_o, ref_lse, _ = _ref(qkv, B, N, h, hd)
    err = (out.double().cpu() - ref_o).abs().max().item()
    print(f"fwd N={N} hd={hd}: max-abs {err:.3e} (|o| max {ref_o.abs().max():.2f})")
    assert err < 2e-2                      # P and O are rounded to bfloat16 (2^-9 relative), |o| <~ 3
    assert (lse.double().cpu().reshape(B, h, N) - ref_lse).abs().max() < 2e-3
    out2, _ = _fwd(qkv.cuda(), B, N, h, hd, want_lse=False)     # eval mode: no lse2 buffer
    assert torch.equal(out, out2)


def test_forward_online_softmax_rescale_branch():
    """keys of growing score along the sequence: the running maximum moves at EVERY 64-key tile, so every tile's
    rescale of the accumulated O / l is exercised (a stale factor would be off by orders of magnitude)"""
    B, N, h, hd = 1, 512, 2, 128
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(B, N, 3, h, hd, generator=g) * 0.3
    qkv[:, :, 0, :, 0] = 6.0                                     # every query looks along feature 0
    qkv[:, :, 1, :, 0] = torch.linspace(-20, 20, N)[None, :, None]   # ... where the keys grow: scores span ~ +-10
    d = qkv.to(torch.bfloat16).reshape(B * N, 3 * h * hd)
    out, lse = _fwd(d.cuda(), B, N, h, hd)
    ref_o, ref_lse, _ = _ref(d, B, N, h, hd)
    assert (out.double().cpu() - ref_o).abs().max() < 2e-2
    assert (lse.double().cpu().reshape(B, h, N) - ref_lse).abs().max() < 5e-3


@pytest.mark.parametrize("B,N,h,hd", SHAPES)
def test_backward_random_matches_float64_autograd(B, N, h, hd):
    g = torch.Generator().manual_seed(11 * N + hd)
    qkv = (torch.randn(B * N, 3 * h * hd, generator=g) * 1.2).to(torch.bfloat16)
    dout = torch.randn(B * N, h * hd, generator=g).to(torch.bfloat16)
    out, lse = _fwd(qkv.cuda(), B, N, h, hd)
    dqkv, delta = _bwd(qkv.cuda(), out, dout.cuda(), lse, B, N, h, hd)
    ref_o, _, ref_d = _ref(qkv, B, N, h, hd, dout)
    assert not torch.isnan(dqkv.float()).any()
    want_delta = (dout.double() * ref_o).reshape(B, N, h, hd).sum(-1).permute(0, 2, 1).reshape(B * h, N)
    assert (delta.double().cpu() - want_delta).abs().max() < 3e-2
    D = h * hd
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        got, ref = dqkv[:, sl].double().cpu(), ref_d[:, sl]
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        cos = float((got.flatten() @ ref.flatten()) / (got.norm() * ref.norm()))
        print(f"bwd N={N} hd={hd} {name}: max-abs/max {err:.3e} cosine {cos:.6f}")
        assert err < 2.5e-2 and cos > 0.9995, (name, err, cos)


def test_backward_structured_exact_dv():
    """uniform attention (q = 0 -> P = 1/N exactly, a power of two): dV[k] = (1/N) sum_q dO[q] must be exact for integer
    dO; dQ and dK vanish because dS = P (dP - delta) is a pure rounding residue times k = 0 / q = 0"""
    B, N, h, hd = 1, 256, 2, 128
    g = torch.Generator().manual_seed(3)
    qkv = torch.zeros(B, N, 3, h, hd)
    qkv[:, :, 2] = torch.randint(-4, 5, (B, N, h, hd), generator=g).float()
    dout = torch.randint(-4, 5, (B * N, h * hd), generator=g).float()
    d = qkv.to(torch.bfloat16).reshape(B * N, 3 * h * hd).cuda()
    out, lse = _fwd(d, B, N, h, hd)
    dqkv, _ = _bwd(d, out, dout.to(torch.bfloat16).cuda(), lse, B, N, h, hd)
    D = h * hd
    want_dv = (dout.sum(0) / N).to(torch.bfloat16).float()[None, :].expand(B * N, D)
    assert torch.equal(dqkv[:, 2 * D:].float().cpu(), want_dv)
    assert float(dqkv[:, :2 * D].float().abs().max()) == 0.0


@pytest.mark.parametrize("cfgname", ["tiny", "d256"])
def test_model_fused_vs_unfused_attention_bf16(cfgname):
    """whole bf16 model with the fused kernels against the same model on the batched-GEMM attention path: logits and
    all gradients agree to bf16 rounding (both round P / dS to bfloat16 once)"""
    from functools import partial
    import torch.nn as nn
    import htrvt_amd
    from htrvt_amd.model import HTR_VT
    from oracle import htrvt_oracle as O
    D, depth, heads = (64, 2, 2) if cfgname == "tiny" else (256, 4, 4)
    cfg = O.Config(80, (64, 512), embed_dim=D, depth=depth, num_heads=heads)
    sd = O.init_state_dict(cfg, seed=7, randomize_affine=True)
    x, targets, lengths = O.synthetic_batch(4, 64, 512, 80, cfg.num_patches, seed=3)
    torch.manual_seed(5)
    keep = O.span_mask(cfg.num_patches, 0.4, 8)
    res = []
    for fused in (True, False):
        m = HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch, embed_dim=cfg.D,
                                        depth=cfg.depth, num_heads=cfg.heads, mlp_ratio=4,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6), compute_dtype=torch.bfloat16)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        m._engine(torch.device("cuda", torch.cuda.current_device())).fused_attention = fused
        y = m(x.cuda(), keep_mask=keep)
        loss = htrvt_amd.ctc_loss(y, targets, lengths)
        loss.backward()
        res.append((y.detach().float().cpu(), float(loss), {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}))
    (y1, l1, g1), (y0, l0, g0) = res
    assert (y1 - y0).abs().max() < 6e-2 and abs(l1 - l0) < 2e-2 * abs(l0)
    worst = 1.0
    for n in g0:
        if g0[n].numel() < 64 or n.endswith("attn.qkv.bias"):
            continue
        cos = float((g1[n].flatten().double() @ g0[n].flatten().double()) / (g1[n].norm().double() * g0[n].norm().double() + 1e-30))
        worst = min(worst, cos)
        assert cos > 0.97, (n, cos)
    print(cfgname, "fused vs unfused attention: worst gradient cosine", worst)
