"""The fused stem forward (conv1 -> BatchNorm -> ReLU -> max-pool in one pass over the image, batch statistics from the
image's second moments: csrc/stem.hip, htrvt_stem_stats / htrvt_stem_fwd) against the two-kernel form that materialises
the conv1 tensor (htrvt_conv1_fwd + htrvt_bn_finalize + htrvt_bn_relu_maxpool), through the Engine, and against the
float64 oracle (reference: model_v1/model/resnet18.py:74-77, HTR_VT.py:224)."""
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import htrvt_oracle as O

pytestmark = pytest.mark.gpu


def _engine(cfg, sd, dtype):
    from htrvt_amd.model import HTR_VT
    m = HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch, embed_dim=cfg.D,
                                    depth=cfg.depth, num_heads=cfg.heads, mlp_ratio=4,
                                    norm_layer=partial(nn.LayerNorm, eps=1e-6), compute_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    return m, m._engine(torch.device("cuda:0")), dict(m.state_dict(keep_vars=True))


def _stem(cfg, sd, x, dtype, fused, train=True):
    m, eng, P = _engine(cfg, sd, dtype)
    eng.fuse_stem_forward = fused
    with torch.no_grad():
        eng.forward(P, x, train=train, save=True)
    sv = eng.saved
    out = dict(a=sv["stem_blocks"][0]["x"].float().cpu(), idx=sv["idx"].cpu(), bn=[t.float().cpu() for t in sv["bn1"] if t is not None],
               rm=P["patch_embed.bn1.running_mean"].detach().cpu().clone(), rv=P["patch_embed.bn1.running_var"].detach().cpu().clone(),
               c1=sv["c1"])
    eng.saved = None
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("W,D,u8", [(512, 64, False), (576, 256, False), (512, 768, True), (2048, 64, False)])
def test_fused_stem_matches_the_two_kernel_form(dtype, W, D, u8):
    cfg = O.Config(80, (64, W), embed_dim=D, depth=1, num_heads=2 if D == 64 else 4)
    sd = O.init_state_dict(cfg, seed=5, randomize_affine=True)
    g = torch.Generator().manual_seed(W + D)
    x = torch.rand(3, 1, 64, W, generator=g)
    if u8:
        x = (x * 255).round().to(torch.uint8)
    x = x.cuda()
    ref = _stem(cfg, sd, x, dtype, fused=False)
    got = _stem(cfg, sd, x, dtype, fused=True)
    assert ref["c1"] is not None and got["c1"] is None          # the fused form never materialises conv1
    # batch statistics: analytic (image moments, double) vs summed over the float32 conv outputs
    for a, b in zip(got["bn"], ref["bn"]):
        assert torch.allclose(a, b, rtol=2e-5, atol=2e-6)
    assert torch.allclose(got["rm"], ref["rm"], rtol=1e-5, atol=1e-7) and torch.allclose(got["rv"], ref["rv"], rtol=2e-5, atol=1e-7)
    # pooled activations: float32 differs only through the statistics' rounding; the bfloat16 two-kernel form rounds the
    # conv1 tensor to bfloat16 BEFORE the BatchNorm, the fused form does not (one bf16 ulp of the activation)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    scale = ref["a"].abs().max().item()
    assert (got["a"] - ref["a"]).abs().max().item() <= tol * scale
    same = (got["idx"] == ref["idx"]).float().mean().item()
    assert same > (0.9999 if dtype == torch.float32 else 0.97), same


def test_fused_stem_against_the_float64_oracle():
    """pooled stem output of the fused kernels vs the oracle's float64 conv1 -> BN(train) -> ReLU -> max_pool2d"""
    import torch.nn.functional as F
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=1, num_heads=2)
    sd = O.init_state_dict(cfg, seed=9, randomize_affine=True)
    x = torch.rand(2, 1, 64, 512, generator=torch.Generator().manual_seed(4))
    got = _stem(cfg, sd, x.cuda(), torch.float32, fused=True)
    xd = x.double()
    xw = F.layer_norm(xd, xd.shape[1:], eps=1e-5)            # HTR_VT.py:134-136,224: param-free LayerNorm over (C,H,W)
    y = F.conv2d(xw, sd["patch_embed.conv1.weight"].double(), stride=(2, 1), padding=1)
    y = F.batch_norm(y, None, None, sd["patch_embed.bn1.weight"].double(), sd["patch_embed.bn1.bias"].double(), True, 0.1, 1e-5)
    ref = F.max_pool2d(F.relu(y), 3, stride=(2, 1), padding=1).permute(0, 2, 3, 1)
    assert (got["a"].double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("W,D,u8", [(576, 256, False), (1024, 768, True), (64, 128, False)])
def test_mfma_stem_against_the_float64_oracle(W, D, u8):
    """bf16 path, C1 = D/4 a multiple of 32: conv1 runs as a float16 MFMA product with the pooling done in the accumulator
    layout (csrc/stem_mfma.hip).  Against the oracle's float64 conv1 -> BN(train) -> ReLU -> max_pool2d: values within a
    bf16 ulp of the largest activation, arg-max bytes equal wherever the float64 window maximum is not a near tie."""
    import torch.nn.functional as F
    cfg = O.Config(80, (64, W), embed_dim=D, depth=1, num_heads=4)
    sd = O.init_state_dict(cfg, seed=11, randomize_affine=True)
    x = torch.rand(2, 1, 64, W, generator=torch.Generator().manual_seed(W))
    if u8:
        x = (x * 255).round().to(torch.uint8)
    got = _stem(cfg, sd, x.cuda(), torch.bfloat16, fused=True)
    xd = (x.double() / 255.0) if u8 else x.double()
    xw = F.layer_norm(xd, xd.shape[1:], eps=1e-5)
    y = F.conv2d(xw, sd["patch_embed.conv1.weight"].double(), stride=(2, 1), padding=1)
    y = F.batch_norm(y, None, None, sd["patch_embed.bn1.weight"].double(), sd["patch_embed.bn1.bias"].double(), True, 0.1, 1e-5)
    ref, ridx = F.max_pool2d(F.relu(y), 3, stride=(2, 1), padding=1, return_indices=True)
    ref = ref.permute(0, 2, 3, 1)
    a = got["a"].double()
    assert a.shape == ref.shape
    assert (a - ref).abs().max().item() < 2.0 ** -7 * ref.abs().max().item(), (a - ref).abs().max().item()
    # arg-max byte = 3 * window row + window column (15: closed ReLU); compare where the oracle's maximum is open
    Hc = y.shape[2]
    rr, cc = ridx // W, ridx % W                                 # conv row / column of the arg-max
    ph = torch.arange(ref.shape[1]).view(1, 1, -1, 1)
    pw = torch.arange(W).view(1, 1, 1, -1)
    want = (3 * (rr - (2 * ph - 1)) + (cc - (pw - 1))).permute(0, 2, 3, 1)
    open_ = ref > 1e-2 * ref.abs().max()
    same = (got["idx"].long()[open_] == want[open_]).float().mean().item()
    assert same > 0.97, same
    assert (got["idx"][ref == 0] == 15).float().mean().item() > 0.97


def test_fused_stem_eval_mode_uses_running_statistics():
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=1, num_heads=2)
    sd = O.init_state_dict(cfg, seed=5, randomize_affine=True)
    x = torch.rand(2, 1, 64, 512, generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for fused in (False, True):
        m, eng, P = _engine(cfg, sd, torch.float32)
        m.eval()
        eng.fuse_stem_forward = fused
        with torch.no_grad():
            outs.append(eng.forward(P, x, train=False, save=False).cpu())
    assert (outs[0] - outs[1]).abs().max().item() < 1e-5


@pytest.mark.parametrize("u8", [False, True])
def test_fused_stem_statistics_on_smooth_padded_lines(u8):
    """Real line scans are smooth strokes beside large areas padded with 1.0 (data/dataset.py:129-130), not white noise:
    the image's second-moment matrix R is then far from diagonal, and for an edge (zero-sum) conv1 filter the batch
    statistic sum y^2 = w^T R w is a difference of large, nearly equal entries.  Mean / rstd of the fused form
    (htrvt_stem_stats) against the float64 evaluation of conv1 over the whitened image."""
    import torch.nn.functional as F
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=1, num_heads=2)
    sd = O.init_state_dict(cfg, seed=9, randomize_affine=True)
    w = sd["patch_embed.conv1.weight"]
    w[::2] -= w[::2].mean(dim=(1, 2, 3), keepdim=True)          # every other filter sums to zero: a pure edge detector
    g = torch.Generator().manual_seed(12)
    x = torch.rand(3, 1, 64, 512, generator=g)
    for _ in range(3):                                           # low-pass: smooth strokes
        x = F.avg_pool2d(F.pad(x, (4, 4, 4, 4), mode="replicate"), 9, stride=1)
    x = (x - x.amin()) / (x.amax() - x.amin())
    for b, cut in enumerate((200, 330, 512)):                    # ragged right padding with 1.0, one line unpadded
        x[b, :, :, cut:] = 1.0
    if u8:
        x = (x * 255).round().to(torch.uint8)
    got = _stem(cfg, sd, x.cuda(), torch.float32, fused=True)
    xd = (x.double() / 255.0) if u8 else x.double()
    if u8:
        xd = (x.float() / 255.0).double()                        # the kernels read uint8 as float32 value / 255
    xw = F.layer_norm(xd, xd.shape[1:], eps=1e-5)
    y = F.conv2d(xw, w.double(), stride=(2, 1), padding=1)
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    _sc, _sf, gmean, grstd = got["bn"]
    sd_ = var.sqrt()
    assert ((gmean.double() - mean).abs() / (sd_ + 1e-3)).max().item() < 1e-4, ((gmean.double() - mean).abs() / (sd_ + 1e-3)).max().item()
    assert ((grstd.double() - rstd).abs() / rstd).max().item() < 2e-4, ((grstd.double() - rstd).abs() / rstd).max().item()
