"""Run-to-run bitwise determinism of the float32 parity path (the stand-in for a race detector on this
hardware) and the eval-mode backward.

The float32 path has no float atomics: split-K weight gradients go through per-K-range slabs summed in K order
(HtrvtGemmDesc.splitk_ws), column sums (bias / LayerNorm / BatchNorm / mask-token gradients) through a two-stage sum,
the CTC occupancy through fixed-order per-class sums.  Two runs from the same inputs must agree in every bit; a data
race or an order-dependent reduction shows up here long before it moves a tolerance-based parity test."""
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import htrvt_oracle as O

pytestmark = pytest.mark.gpu


def _build(cfg, seed, dtype=torch.float32):
    from htrvt_amd.model import HTR_VT
    sd = O.init_state_dict(cfg, seed=seed, randomize_affine=True)
    m = HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch, embed_dim=cfg.D,
                                    depth=cfg.depth, num_heads=cfg.heads, mlp_ratio=4,
                                    norm_layer=partial(nn.LayerNorm, eps=1e-6), compute_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    return sd, m.cuda().train()


def _one_run(cfg, B, steps=2, dtype=torch.float32):
    """`steps` full SAM(AdamW) iterations from a fixed start; returns every gradient of the last pass and every
    parameter / buffer afterwards"""
    from htrvt_amd.trainer import Trainer
    _, m = _build(cfg, 7, dtype=dtype)
    tr = Trainer(m, max_lr=1e-3, betas=(0.9, 0.99), weight_decay=0.5)
    x, targets, lengths = O.synthetic_batch(B, cfg.H, cfg.W, cfg.nb_cls, cfg.num_patches, seed=3)
    xd = x.cuda()
    for it in range(steps):
        masks = []
        for s in (100 + 2 * it, 101 + 2 * it):
            torch.manual_seed(s)
            masks.append(m.generate_span_mask(cfg.num_patches, 0.4, 8))
        loss = tr.sam_step(xd, targets, lengths, masks[0], masks[1], lr=1e-3, rho=0.05)
    torch.cuda.synchronize()
    return float(loss), tr.flat.flat_g.clone(), {k: v.detach().clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("shape", [(64, 2, 2, 512, 4), (256, 4, 4, 512, 8)])
def test_f32_training_iterations_bitwise_reproducible(shape):
    """tiny model, and d256/4L/4h at B=8 (BASELINE config 1's shape) where the conv weight gradients really split K
    (32768 output pixels per launch) and every column sum takes its multi-split path"""
    D, depth, heads, W, B = shape
    cfg = O.Config(80, (64, W), embed_dim=D, depth=depth, num_heads=heads)
    l1, g1, s1 = _one_run(cfg, B)
    l2, g2, s2 = _one_run(cfg, B)
    assert l1 == l2
    assert torch.equal(g1, g2), int((g1 != g2).sum())
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k


def test_bf16_training_iterations_bitwise_reproducible():
    """round 3: the bf16 throughput path splits K through ordered slabs by default (Engine.deterministic), every other
    reduction was already order-fixed: two SAM(AdamW) iterations at d256 / B = 8 repeat bit for bit"""
    cfg = O.Config(80, (64, 512), embed_dim=256, depth=4, num_heads=4)
    l1, g1, s1 = _one_run(cfg, 8, dtype=torch.bfloat16)
    l2, g2, s2 = _one_run(cfg, 8, dtype=torch.bfloat16)
    assert l1 == l2
    assert torch.equal(g1, g2), int((g1 != g2).sum())
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k


@pytest.mark.parametrize("flag", ["table_relayout", "merge_bn_backward", "relu_mask_from_bn"])
def test_launch_merges_are_bit_neutral_bf16(flag):
    """bf16 engine, two AdamW steps with slab split-K, every gradient and parameter bit-identical with and without
    * table_relayout: the one-launch weight re-pack / gradient unpack (csrc/relayout.hip) against per-tensor launches,
    * merge_bn_backward: bn2 + downsample-BN backward of a stage's first block in one pass over the shared gradient,
    * relu_mask_from_bn: conv2's dgrad epilogue rebuilds the ReLU mask from bn1's input instead of reading the activation."""
    from htrvt_amd.trainer import Trainer
    cfg = O.Config(80, (64, 512), embed_dim=256, depth=4, num_heads=4)
    x, targets, lengths = O.synthetic_batch(8, cfg.H, cfg.W, cfg.nb_cls, cfg.num_patches, seed=3)
    outs = []
    for on in (True, False):
        _, m = _build(cfg, 7, dtype=torch.bfloat16)
        tr = Trainer(m, max_lr=1e-3, betas=(0.9, 0.99), weight_decay=0.5)
        tr.engine.deterministic = True
        assert getattr(tr.engine, flag) is True          # the default is the merged form
        setattr(tr.engine, flag, on)
        for it in range(2):
            torch.manual_seed(50 + it)
            mask = m.generate_span_mask(cfg.num_patches, 0.4, 8)
            loss = tr.step(x.cuda(), targets, lengths, mask, lr=1e-3)
        torch.cuda.synchronize()
        outs.append((float(loss), tr.flat.flat_g.clone(), tr.flat.flat_p.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]), int((outs[0][1] != outs[1][1]).sum())
    assert torch.equal(outs[0][2], outs[1][2])


def test_ctc_gradient_bitwise_reproducible():
    from htrvt_amd.ctc import ctc_forward_backward
    rng = np.random.default_rng(2)
    B, T, C = 16, 256, 80
    logits = torch.from_numpy(rng.standard_normal((B, T, C)).astype(np.float32)).cuda()
    lengths = rng.integers(5, 100, size=B).astype(np.int32)
    targets = rng.integers(1, 6, size=int(lengths.sum())).astype(np.int32)      # 5 symbols only: long per-class lists, repeats
    outs = [ctc_forward_backward(logits, targets, lengths) for _ in range(3)]
    for nll, grad in outs[1:]:
        assert torch.equal(nll, outs[0][0]) and torch.equal(grad, outs[0][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_mode_backward_matches_oracle(dtype):
    """model.eval(); loss.backward() (frozen-BatchNorm fine-tuning, saliency): BatchNorm runs on its running statistics
    in the forward and passes dx = dy * gamma / sqrt(running_var + eps) in the backward; dgamma / dbeta follow the same
    constants.  Against torch autograd over the oracle graph in float64."""
    import htrvt_amd
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    sd, m = _build(cfg, 7, dtype)
    # running statistics away from their initial 0 / 1 so that the test sees them
    gen = torch.Generator().manual_seed(4)
    for k, v in sd.items():
        if k.endswith("running_mean"):
            v.copy_(torch.randn(v.shape, generator=gen) * 0.3)
        if k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=gen) * 1.5 + 0.5)
    m.load_state_dict(sd, strict=True)
    m.eval()
    x, targets, lengths = O.synthetic_batch(2, 64, 512, 80, cfg.num_patches, seed=3)
    loss_ref, logits_ref, grads_ref, _ = O.loss_and_grads(sd, cfg, x, targets, lengths, keep_mask=None, train=False,
                                                          dtype=torch.float64)
    xd = x.cuda().requires_grad_(False)
    y = m(xd)
    loss = htrvt_amd.ctc_loss(y, targets, lengths)
    loss.backward()
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    assert (y.detach().cpu().double() - logits_ref).abs().max().item() < (1e-3 if f32 else 0.25)
    assert abs(loss.item() - loss_ref) < (1e-4 if f32 else 5e-2) * abs(loss_ref)
    for k, v in m.state_dict().items():          # an eval-mode pass must not touch the running statistics
        if "running_" in k or k.endswith("num_batches_tracked"):
            assert torch.equal(v.cpu(), sd[k]), k
    for n, p in m.named_parameters():
        if n not in grads_ref:
            continue
        ref = grads_ref[n]
        got = p.grad.cpu().double()
        if f32:
            e = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
            assert e < 2e-2, (n, e)
        else:
            cos = float((got.flatten() @ ref.flatten()) / (got.norm() * ref.norm() + 1e-30))
            assert cos > 0.8, (n, cos)
