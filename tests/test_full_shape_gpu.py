"""Parity at the BENCH shapes of BASELINE.json: configs 2 / 3 = the headline, HTR-VT base (d768 / 4L / 6h, nb_cls 80), 64x1024
lines, batch 128; config 4 = the long line, 64x2048 (N = 512 tokens), batch 64; config 5 = d512 / 12L / 8h, nb_cls 90, batch 32
(the per-GPU share of 256 on 8 GPUs) -- the shapes bench.py and profiles/r0x_configs.md report throughput for.  Kernel variants
are picked per shape (gemm_dma.hip pick_bn /
use_loader_waves, the persistent 8-phase walk, engine._split_k incl. the XCD-grouped multiples of 8, parity-class dgrad
launches), so this is the only place the bench's hot variants meet the oracle end to end.

  * float32 path AND split-bf16 parity path (float32 activations, hi + lo bf16 operands on the bf16 matrix cores):
    eval logits vs the oracle on an 8-image subset (eval-mode BatchNorm makes samples independent; every launch still
    runs at B = 128), train-mode logits (batch statistics over all 128 images, span mask on) and CTC loss vs the oracle's
    full-batch forward: <= 1e-3 on the logits (BASELINE.json north_star), 1e-5 relative on the loss;
  * bfloat16 (the measured path): gated against what the REFERENCE ARITHMETIC ITSELF loses in bfloat16 -- the oracle
    under torch.autocast(bfloat16) -- in relative L2 norm, eval and train (profiles/r04_bf16_ladder.md: the train-mode
    error enters through the 16 batch-statistics BatchNorms, in the reference exactly as here);
  * one full fwd + CTC + bwd of all three paths, gradient by gradient against torch autograd over the oracle on the full
    batch (needs ~45 GB of host memory: skipped on smaller hosts)."""
import os
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import htrvt_oracle as O

pytestmark = pytest.mark.gpu

SPLIT = "split_bf16"
# tag -> (batch, width, nb_cls, embed_dim, depth, heads, images of the eval subset)
CONFIGS = {"cfg2": (128, 1024, 80, 768, 4, 6, 8), "cfg4": (64, 2048, 80, 768, 4, 6, 4), "cfg5": (32, 1024, 90, 512, 12, 8, 4)}


def _host_gib():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


@pytest.fixture(scope="module", params=list(CONFIGS))
def setup(request):
    tag = request.param
    B, W, NB_CLS, D, depth, heads, nsub = CONFIGS[tag]
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    cfg = O.Config(NB_CLS, (64, W), embed_dim=D, depth=depth, num_heads=heads)
    sd = O.init_state_dict(cfg, seed=123, randomize_affine=True)
    x, targets, lengths = O.synthetic_batch(B, 64, W, NB_CLS, cfg.num_patches, seed=0)
    torch.manual_seed(7)
    keep = O.span_mask(cfg.num_patches, 0.4, 8)
    with torch.no_grad():
        sub = torch.arange(0, B, B // nsub)                            # images spread over the batch
        ref_eval = O.forward(sd, cfg, x[sub], train=False)
        ref_train = O.forward(sd, cfg, x, keep_mask=keep, train=True)
        # the reference arithmetic with bfloat16 operands: what bf16 costs the reference itself
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ac_eval = O.forward(sd, cfg, x[sub], train=False).float()
            ac_train = O.forward(sd, cfg, x, keep_mask=keep, train=True).float()
    lp = ref_train.double().permute(1, 0, 2).log_softmax(2)             # compute_loss (train.py:21-30) in float64
    ref_nll = torch.nn.functional.ctc_loss(lp, torch.from_numpy(targets), torch.full((B,), lp.shape[0], dtype=torch.int32),
                                           torch.from_numpy(lengths), blank=0, reduction="none", zero_infinity=True).numpy()
    return dict(tag=tag, B=B, cfg=cfg, sd=sd, x=x, targets=targets, lengths=lengths, keep=keep, sub=sub, ref_eval=ref_eval,
                ref_train=ref_train, ref_nll=np.asarray(ref_nll), ac_eval=ac_eval, ac_train=ac_train, grads={})


def _model(cfg, sd, dtype):
    from htrvt_amd.model import HTR_VT
    m = HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch, embed_dim=cfg.D,
                                    depth=cfg.depth, num_heads=cfg.heads, mlp_ratio=4,
                                    norm_layer=partial(nn.LayerNorm, eps=1e-6), compute_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("dtype", [torch.float32, SPLIT, torch.bfloat16])
def test_bench_shape_eval_and_train_forward(setup, dtype):
    from htrvt_amd.ctc import ctc_forward_backward
    s = setup
    m = _model(s["cfg"], s["sd"], dtype)
    xd = s["x"].cuda()
    m.eval()
    with torch.no_grad():
        y = m(xd)
    y_eval = y[s["sub"].cuda()].cpu()
    err_eval = (y_eval - s["ref_eval"]).abs().max().item()
    m.train()
    with torch.no_grad():
        yt = m(xd, keep_mask=s["keep"])
        nll, _ = ctc_forward_backward(yt, s["targets"], s["lengths"], want_grad=False)
    yt = yt.cpu()
    err_train = (yt - s["ref_train"]).abs().max().item()
    rel_loss = abs(float(nll.mean()) - float(s["ref_nll"].mean())) / abs(float(s["ref_nll"].mean()))
    agree = (yt.argmax(-1) == s["ref_train"].argmax(-1)).float().mean().item()
    l2_eval, l2_train = _rel_l2(y_eval, s["ref_eval"]), _rel_l2(yt, s["ref_train"])
    print(f"{s['tag']} {dtype} B={s['B']} 64x{s['cfg'].W} d{s['cfg'].D}/{s['cfg'].depth}L: eval logits max-abs {err_eval:.3e} rel-L2 {l2_eval:.3e}, train logits max-abs {err_train:.3e} "
          f"rel-L2 {l2_train:.3e}, CTC loss rel {rel_loss:.3e}, arg-max agreement {agree:.4f}")
    if dtype != torch.bfloat16:
        assert err_eval < 1e-3 and err_train < 1e-3
        assert rel_loss < 1e-5
    else:
        ac_eval, ac_train = _rel_l2(s["ac_eval"], s["ref_eval"]), _rel_l2(s["ac_train"], s["ref_train"])
        ac_agree = (s["ac_train"].argmax(-1) == s["ref_train"].argmax(-1)).float().mean().item()
        print(f"   the oracle under bf16 autocast: eval rel-L2 {ac_eval:.3e}, train rel-L2 {ac_train:.3e} (max-abs "
              f"{(s['ac_train'] - s['ref_train']).abs().max().item():.3e}), arg-max agreement {ac_agree:.4f}")
        # (eval: 6.7e-3 here vs 5.3e-3 for the 8-image autocast subset -- the GPU figure is over the same 8 images; train: 2.6e-2 vs 3.3e-2)
        assert l2_eval < 1.5 * ac_eval and l2_train < 1.25 * ac_train, (l2_eval, ac_eval, l2_train, ac_train)
        assert agree > ac_agree - 0.01 and rel_loss < 2e-2
        assert agree > 0.95                              # absolute floor beside the gate relative to the autocast oracle
        assert err_eval < 6e-2 and err_train < 0.3


def _gpu_grads(s):
    """one fwd + fused CTC + bwd per path at the bench shape, cached for the two tests below"""
    import htrvt_amd
    from _decisions import decision_flips
    if not s["grads"]:
        sv32 = None
        for dtype in (torch.float32, SPLIT, "hybrid", torch.bfloat16):
            m = _model(s["cfg"], s["sd"], SPLIT if dtype == "hybrid" else dtype).train()
            y = m(s["x"].cuda(), keep_mask=s["keep"])
            if dtype == torch.float32:
                sv32 = y.grad_fn.saved_acts          # kept past this path's backward: the float32 path's forward state
            elif dtype == SPLIT:
                s["flips"] = decision_flips(sv32, y.grad_fn.saved_acts)
            elif dtype == "hybrid":                  # the split-bf16 BACKWARD over the float32 path's forward state
                y.grad_fn.saved_acts, sv32 = sv32, None
            loss = htrvt_amd.ctc_loss(y, s["targets"], s["lengths"])
            loss.backward()
            s["grads"][dtype] = ({n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None}, float(loss))
            del m, y, loss
            torch.cuda.empty_cache()
    return s["grads"]


def test_bench_shape_training_step_gradients_across_paths(setup):
    """one fwd + CTC + bwd at the bench shape on all three paths, path against path; bf16 by cosine.  float32 against
    split-bf16, measured instead of asserted: (i) the split path's BACKWARD over the float32 path's forward
    state (same kernels and hi + lo products, only the saved activations / decisions are the float32 path's) is within
    1.5e-3 relative L2 of the float32 path on every tensor (measured 2.3e-4 ... 5.7e-4); (ii) the two forward passes differ in
    a counted handful of discrete decisions (tests/_decisions.py: 9 767 of 1.84e9 ReLU signs / arg-max positions at the
    headline shape); (iii) the pure split path is within 3e-2 (measured 1.30e-2 = those decisions, not arithmetic).
    Exercises split-K wgrad, the fused dgrad epilogues, the fused attention backward, the merged strided dgrad launches,
    the streaming 1x1 kernel and the persistent Linear kernels at exactly the bench's shapes."""
    s = setup
    grads = _gpu_grads(s)
    g32, l32 = grads[torch.float32]
    gsp, lsp = grads[SPLIT]
    g16, l16 = grads[torch.bfloat16]
    ghy, _ = grads["hybrid"]
    ref_loss = float(s["ref_nll"].mean())
    assert abs(l32 - ref_loss) < 1e-5 * abs(l32) and abs(lsp - ref_loss) < 1e-5 * abs(lsp)
    assert abs(l16 - l32) < 2e-2 * abs(l32)
    worst, worst_sp, worst_hy = (1.0, None), (0.0, None), (0.0, None)
    for n in g32:
        if g32[n].numel() < 64 or n.endswith("attn.qkv.bias"):
            continue
        a, b_ = g16[n].flatten().double(), g32[n].flatten().double()
        cos = float(a @ b_ / (a.norm() * b_.norm() + 1e-30))
        worst = min(worst, (cos, n))
        assert cos > 0.9, (n, cos)
        e = float((gsp[n].flatten().double() - b_).norm() / (b_.norm() + 1e-30))
        worst_sp = max(worst_sp, (e, n))
        assert e < 3e-2, ("split vs float32", n, e)      # measured <= 1.30e-2 (cfg2 / cfg4 / cfg5); explained below: decisions, not arithmetic
        # the split-bf16 backward over the float32 path's forward state: what is left is the hi + lo operand rounding of the
        # backward GEMMs alone -- every tensor at rounding level.  What the pure split path shows beyond this is forward
        # decisions (ReLU signs, max-pool arg-max) that fell the other way, counted below.
        eh = float((ghy[n].flatten().double() - b_).norm() / (b_.norm() + 1e-30))
        worst_hy = max(worst_hy, (eh, n))
        assert eh < 1.5e-3, ("split backward over the float32 forward state vs float32", n, eh)      # measured 2.3e-4 ... 5.7e-4
    nflip = sum(v[0] for v in s["flips"].values())
    nall = sum(v[1] for v in s["flips"].values())
    print(f"{s['tag']} B={s['B']} bf16 vs f32 training-step gradients: worst cosine", worst)
    print(f"{s['tag']} B={s['B']} split-bf16 vs float32 gradients: worst rel-L2 {worst_sp}; with the float32 path's forward decisions {worst_hy}; "
          f"decisions that differ between the two forward passes: {nflip} of {nall}", {k: v[0] for k, v in s["flips"].items() if v[0]})
    if worst_sp[0] > 1e-3:
        assert nflip > 0


def test_bench_shape_training_step_gradients_vs_oracle(setup):
    """the same gradients, tensor by tensor against torch autograd over the CPU restatement on the FULL batch (float32,
    1-2 minutes and ~45 GB of host memory at the headline shape: a test of its own, so that a small host shows up as ONE
    skipped test instead of hiding the path-against-path checks above)."""
    s = setup
    need = 60 if s["tag"] != "cfg5" else 24
    if _host_gib() < need:
        pytest.skip(f"oracle backward on the full batch needs ~{need - 15} GB of host memory ({_host_gib():.0f} GiB available)")
    grads = _gpu_grads(s)
    g32, l32 = grads[torch.float32]
    gsp, lsp = grads[SPLIT]
    g16, l16 = grads[torch.bfloat16]
    import time
    t0 = time.time()
    loss_ref, _, gref, _ = O.loss_and_grads(s["sd"], s["cfg"], s["x"], s["targets"], s["lengths"], keep_mask=s["keep"], train=True)
    print(f"{s['tag']} oracle fwd+bwd on the full batch: {time.time() - t0:.0f} s, loss {loss_ref:.6f} (GPU float32 {l32:.6f}, split-bf16 {lsp:.6f})")
    assert abs(l32 - loss_ref) < 1e-5 * abs(loss_ref) and abs(lsp - loss_ref) < 1e-5 * abs(loss_ref)
    worst32, worstsp, worst16 = (0.0, None), (0.0, None), (1.0, None)
    for n, r in gref.items():
        if n not in g32 or r.numel() < 64 or n.endswith("attn.qkv.bias"):
            continue
        r = r.flatten().double()
        a32, asp, a16 = g32[n].flatten().double(), gsp[n].flatten().double(), g16[n].flatten().double()
        e32 = float((a32 - r).norm() / (r.norm() + 1e-30))
        esp = float((asp - r).norm() / (r.norm() + 1e-30))
        c16 = float(a16 @ r / (a16.norm() * r.norm() + 1e-30))
        worst32, worstsp, worst16 = max(worst32, (e32, n)), max(worstsp, (esp, n)), min(worst16, (c16, n))
        # per-tensor bounds: the encoder / head see no ReLU or arg-max discontinuity (measured 3e-5 / 6.5e-5), the stem does.
        # float32 path: <= 4.8e-3 measured, gate 1.5e-2.  split-bf16 path: <= 1.30e-2 measured, gate 3e-2 -- DERIVED, not widened:
        # test_bench_shape_training_step_gradients_across_paths runs the split path's backward over the float32 path's forward
        # state and finds every tensor within 5.7e-4 of the float32 path, and counts the forward decisions (ReLU signs, max-pool
        # arg-max) on which the two forward passes differ: 9 767 of 1.84e9 at this shape.  The split path's distance from the
        # oracle is the float32 path's plus those decisions; twice the measured figure is the gate.
        stem = not n.startswith(("blocks.", "head.", "norm."))
        assert e32 < (1.5e-2 if stem else 1e-3), (n, e32)
        assert esp < (3e-2 if stem else 1e-3), (n, esp)
        assert c16 > 0.9, (n, c16)
    last = s["cfg"].depth - 1
    for n in ("head.weight", f"blocks.{last}.mlp.fc1.weight", "blocks.0.attn.qkv.weight", "patch_embed.layer3.1.conv2.weight",
              "patch_embed.layer1.0.conv1.weight", "patch_embed.conv1.weight"):
        r = gref[n].flatten().double()
        print(f"   {n:40s} float32 rel-L2 {float((g32[n].flatten().double() - r).norm() / r.norm()):.3e}   "
              f"split-bf16 rel-L2 {float((gsp[n].flatten().double() - r).norm() / r.norm()):.3e}   "
              f"bf16 cosine {float(g16[n].flatten().double() @ r / (g16[n].flatten().double().norm() * r.norm())):.5f}")
    print(f"{s['tag']} B={s['B']} gradients vs the oracle: float32 worst rel-L2", worst32, "| split-bf16 worst rel-L2", worstsp,
          "| bf16 worst cosine", worst16)
