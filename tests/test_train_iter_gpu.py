"""The training iteration around the path on the GPU: the flat SAM / AdamW kernels one by one
against torch on identical inputs, the SAM iteration's two gradient passes against the float64 oracle evaluated at the
GPU's OWN weights (so a ReLU / arg-max flip in an earlier step cannot leak into the verdict of a later one), the
reference-generated two-iteration trace (tools/make_goldens_train.py), ModelEma (one multi-tensor launch) and the device
greedy CTC decode."""
import math
import os
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import htrvt_oracle as O
from test_train_iter_cpu import check_iterations, f64_trace

pytestmark = pytest.mark.gpu

EPS32 = 2.0 ** -24


def _tiny(dtype=torch.float32, seed=7):
    from htrvt_amd.model import HTR_VT
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    sd = O.init_state_dict(cfg, seed=seed, randomize_affine=True)
    m = HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch, embed_dim=cfg.D,
                                    depth=cfg.depth, num_heads=cfg.heads, mlp_ratio=4,
                                    norm_layer=partial(nn.LayerNorm, eps=1e-6), compute_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    return cfg, sd, m.cuda().train()


def _masks(m, cfg, it):
    out = []
    for s in (100 + 2 * it, 101 + 2 * it):
        torch.manual_seed(s)
        out.append(m.generate_span_mask(cfg.num_patches, 0.4, 8))
    return out


def _sam_scale(nsq, rho):
    """scale = rho / (|g| + 1e-12) in float32 exactly as sam.py:20-21 evaluates it on float32 tensors -- torch's
    `float / tensor` is reciprocal-then-multiply -- with correctly rounded (IEEE) operations: numpy on the host (a GPU
    build of torch may use approximate device division / sqrt, which is not what is under test here)"""
    n = np.sqrt(np.float32(nsq.item())) + np.float32(1e-12)
    return float(np.float32(1.0) / n * np.float32(rho))


# --------------------------------------------------------------------------------------------------------------------
# the flat kernels, one by one, against torch on identical inputs
# --------------------------------------------------------------------------------------------------------------------
def test_sumsq_first_step_restore_match_torch():
    """utils/sam.py:15-38 on a flat buffer: |g|^2 (two-stage sum: compared in double), the climb w + rho g / (|g| +
    1e-12) (each product and sum rounded once, like torch's `p.add_(g * scale)`: bit for bit), the restore (bit for bit)"""
    from htrvt_amd._lib import check, lib
    from htrvt_amd.ops import ptr, stream
    gen = torch.Generator().manual_seed(5)
    n = (1 << 20) + 4 * 37
    p0 = (torch.randn(n, generator=gen) * 0.05).cuda()
    g = (torch.randn(n, generator=gen) * 3e-3).cuda()
    p, old = p0.clone(), torch.empty_like(p0)
    partial_ = torch.empty(lib.htrvt_sumsq_blocks(n), device="cuda")
    nsq = torch.empty(1, device="cuda")
    check(lib.htrvt_sumsq(ptr(g), n, ptr(partial_), ptr(nsq), stream()), "sumsq")
    want = float((g.double() ** 2).sum())
    assert abs(float(nsq) - want) <= 4 * EPS32 * want, (float(nsq), want)
    rho = 0.05
    check(lib.htrvt_sam_first_step(ptr(p), ptr(g), ptr(old), n, rho, ptr(nsq), stream()), "sam_first_step")
    scale = _sam_scale(nsq, rho)
    assert torch.equal(old, p0)
    assert torch.equal(p, p0 + g * scale)         # e_w = g * scale (rounded), p.add_(e_w) (sam.py:24-25)
    check(lib.htrvt_sam_restore(ptr(p), ptr(old), n, stream()), "sam_restore")
    assert torch.equal(p, p0)
    # the two-stage sum is a fixed tree: same bits on every call
    nsq2 = torch.empty(1, device="cuda")
    check(lib.htrvt_sumsq(ptr(g), n, ptr(partial_), ptr(nsq2), stream()), "sumsq")
    assert torch.equal(nsq, nsq2)


def test_adamw_kernel_matches_torch_optim():
    """htrvt_adamw against torch.optim.AdamW (the single-tensor CPU implementation the reference's SAM wraps,
    train.py:94) over three steps with fresh gradients.  Not bit for bit: ATen's vectorised CPU kernels and a scalar
    float32 evaluation of the same expressions already differ by one ulp in ~2 % of the elements (FMA use inside lerp_ /
    addcmul_); parameters and both moments must stay within 4 float32 ulps (the update term counted in ulps of lr)."""
    from htrvt_amd._lib import check, lib
    from htrvt_amd.ops import ptr, stream
    gen = torch.Generator().manual_seed(11)
    n = 1 << 18
    p_ref = nn.Parameter(torch.randn(n, generator=gen) * 0.05)
    opt = torch.optim.AdamW([p_ref], lr=1e-3, betas=(0.9, 0.99), weight_decay=0.5, foreach=False)
    p = p_ref.detach().clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        g = torch.randn(n, generator=gen) * (10.0 ** float(torch.randint(-6, 0, (1,), generator=gen)))
        st = opt.state[p_ref]
        m_prev = st["exp_avg"].clone() if "exp_avg" in st else torch.zeros(n)
        v_prev = st["exp_avg_sq"].clone() if "exp_avg_sq" in st else torch.zeros(n)
        p_ref.grad = g.clone()
        opt.step()
        g_d = g.cuda()   # named: the launch must not read a tensor that has already been freed
        check(lib.htrvt_adamw(ptr(p), ptr(g_d), ptr(m), ptr(v), n, 1e-3, 0.9, 0.99, 1e-8, 0.5, step, stream()), "adamw")
        st = opt.state[p_ref]
        # 4 ulps of the LARGEST quantity entering each update (m + w (g - m) cancels when g ~ -m: the error is an ulp of
        # the operands, not of the small result; the parameter update is of size lr)
        scales = {"p": torch.maximum(p_ref.detach().abs(), torch.full((n,), 1e-3)),
                  "m": torch.maximum(torch.maximum(st["exp_avg"].abs(), m_prev.abs()), g.abs()),
                  "v": torch.maximum(torch.maximum(st["exp_avg_sq"], v_prev), g * g)}
        for name, got, ref in (("p", p, p_ref.detach()), ("m", m, st["exp_avg"]), ("v", v, st["exp_avg_sq"])):
            d = (got.cpu() - ref).abs()
            tol = 8 * EPS32 * scales[name] + 1e-37
            bad = d > tol
            i = int(torch.argmax(d - tol))
            assert not bool(bad.any()), (step, name, int(bad.sum()), float(d[i]), float(ref[i]), float(got.cpu()[i]))


# --------------------------------------------------------------------------------------------------------------------
# one SAM(AdamW) iteration, every stage against the oracle evaluated at the GPU's own weights
# --------------------------------------------------------------------------------------------------------------------
def _grad_check(G, ref, tag, tol=2e-2):
    for n, r in ref.items():
        got = G[n].detach().cpu().double()
        e = (got - r).abs().max().item() / max(r.abs().max().item(), 1e-6)
        assert e < tol, (tag, n, e)


def test_sam_iteration_stages_against_oracle_at_own_weights():
    from htrvt_amd.trainer import Trainer
    cfg, sd, m = _tiny()
    lr, rho = 1e-3, 0.05
    tr = Trainer(m, max_lr=lr, betas=(0.9, 0.99), weight_decay=0.5)
    x, targets, lengths = O.synthetic_batch(4, 64, 512, 80, cfg.num_patches, seed=3)
    xd = x.cuda()
    k1, k2 = _masks(m, cfg, 0)
    names = [n for n, p in m.named_parameters() if p.requires_grad]

    # pass 1 at w
    loss1 = float(tr.forward_backward(xd, targets, lengths, k1))
    ref_loss1, _, ref_g1, _ = O.loss_and_grads(sd, cfg, x, targets, lengths, keep_mask=k1, train=True, dtype=torch.float64)
    assert abs(loss1 - ref_loss1) < 1e-4 * abs(ref_loss1)
    _grad_check(tr.flat.G, ref_g1, "pass1")
    g1 = {n: tr.flat.G[n].detach().clone() for n in names}
    w0 = {n: p.detach().clone() for n, p in m.named_parameters()}

    # climb: w + rho g / (|g| + 1e-12), bit for bit given the kernel's own norm
    tr.sam_first_step(rho)
    nsq = tr._sam_buf[2]
    want_nsq = float(sum((g.double() ** 2).sum() for g in g1.values()))
    assert abs(float(nsq) - want_nsq) <= 8 * EPS32 * want_nsq
    scale = _sam_scale(nsq, rho)
    for n, p in m.named_parameters():
        if n in g1:
            assert torch.equal(p.detach(), w0[n] + g1[n] * scale), n

    # pass 2 at the GPU's own perturbed weights (BatchNorm running statistics do not enter a train-mode forward)
    sd_pert = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    loss2 = float(tr.forward_backward(xd, targets, lengths, k2))
    ref_loss2, _, ref_g2, _ = O.loss_and_grads(sd_pert, cfg, x, targets, lengths, keep_mask=k2, train=True, dtype=torch.float64)
    assert abs(loss2 - ref_loss2) < 1e-4 * abs(ref_loss2)
    _grad_check(tr.flat.G, ref_g2, "pass2")
    g2 = {n: tr.flat.G[n].detach().cpu().clone() for n in names}

    # restore + AdamW with the second gradients: torch.optim.AdamW on the same (w, g2)
    tr.sam_second_step(lr)
    for n in names:
        p_ref = nn.Parameter(w0[n].cpu().clone())
        p_ref.grad = g2[n]
        torch.optim.AdamW([p_ref], lr=lr, betas=(0.9, 0.99), weight_decay=0.5, foreach=False).step()
        got = dict(m.named_parameters())[n].detach().cpu()
        d = (got - p_ref.detach()).abs()
        assert float(d.max()) <= 4 * EPS32 * (float(p_ref.abs().max()) + lr), (n, float(d.max()))


# --------------------------------------------------------------------------------------------------------------------
# the reference's two-iteration trace
# --------------------------------------------------------------------------------------------------------------------
def test_sam_step_and_ema_match_reference_iterations(golden_dir):
    """iteration 0 element by element against the reference's own run; iteration 1 through the direction of the update
    measured against the float64 evaluation of the same two iterations -- the GPU must be as close to it as the
    reference's float32 run is (check_iterations explains why two float32 runs cannot be compared tensor by tensor)."""
    from htrvt_amd.ema import ModelEma
    from htrvt_amd.trainer import Trainer
    g = np.load(os.path.join(golden_dir, "train_iter.npz"))
    cfg, sd, m = _tiny()
    ema = ModelEma(m, 0.9999)
    tr = Trainer(m, max_lr=float(g["lr"]), betas=(0.9, 0.99), weight_decay=0.5)
    x, targets, lengths = O.synthetic_batch(4, 64, 512, 80, cfg.num_patches, seed=3)
    xd = x.cuda()
    res = []
    for it in range(2):
        k1, k2 = _masks(m, cfg, it)
        loss = tr.sam_step(xd, targets, lengths, k1, k2, lr=float(g["lr"]), rho=float(g["rho"]))
        ema.update(m, num_updates=it / 2)
        res.append((float(loss), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
                    {k: v.detach().cpu().clone() for k, v in ema.ema.state_dict().items()}))
    check_iterations(res, g, strict_tol=3e-5, per_tensor_cos_min=None, f64=f64_trace(g), verbose=True)


def test_reference_style_loop_through_nn_module(golden_dir):
    """The reference's own loop shape (train.py:113-128) through the nn.Module surface: torch.optim.AdamW, a SAM wrapper
    that re-binds p.data in second_step (utils/sam.py:31-36), a deepcopy EMA whose update walks state_dict()
    (utils/utils.py:158-173), ATen log_softmax + CTCLoss with cuDNN off.  Pins the engine's packed-weight cache
    invalidation (engine.py `_wkey`): a stale pack would reproduce iteration 0 and miss iteration 1."""
    from copy import deepcopy
    g = np.load(os.path.join(golden_dir, "train_iter.npz"))
    cfg, sd, m = _tiny()
    lr, rho = float(g["lr"]), float(g["rho"])
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=lr, betas=(0.9, 0.99), weight_decay=0.5)
    ema_m = deepcopy(m).eval()
    for p in ema_m.parameters():
        p.requires_grad_(False)
    x, targets, lengths = O.synthetic_batch(4, 64, 512, 80, cfg.num_patches, seed=3)
    xd, tg, tl = x.cuda(), torch.from_numpy(targets).cuda(), torch.from_numpy(lengths).cuda()
    crit = torch.nn.CTCLoss(reduction="none", zero_infinity=True)

    def compute_loss(seed):                                       # train.py:21-30
        torch.manual_seed(seed)                                   # the span mask comes from the CPU generator
        preds = m(xd, 0.4, 8, use_masking=True).float()
        lp = preds.permute(1, 0, 2).log_softmax(2)
        torch.backends.cudnn.enabled = False
        loss = crit(lp, tg, torch.IntTensor([lp.shape[0]] * 4).cuda(), tl).mean()
        torch.backends.cudnn.enabled = True
        return loss

    res = []
    for it in range(2):
        opt.zero_grad()
        loss = compute_loss(100 + 2 * it)
        loss.backward()
        # SAM.first_step (sam.py:15-27)
        norm = torch.norm(torch.stack([p.grad.norm(p=2) for p in params if p.grad is not None]), p=2)
        scale = rho / (norm + 1e-12)
        old = {}
        with torch.no_grad():
            for p in params:
                if p.grad is None:
                    continue
                old[p] = p.data.clone()
                p.add_(p.grad * scale)
        opt.zero_grad()
        compute_loss(101 + 2 * it).backward()
        for p in params:                                          # SAM.second_step (sam.py:29-38): REBINDS the storage
            if p in old:
                p.data = old[p]
        opt.step()
        opt.zero_grad()
        d = min(0.9999, (1 + it / 2) / (10 + it / 2))             # ModelEma.update (utils.py:158-173)
        with torch.no_grad():
            msd = m.state_dict()
            for k, ev in ema_m.state_dict().items():
                ev.copy_(ev * d + (1.0 - d) * msd[k].detach())
        res.append((float(loss), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
                    {k: v.detach().cpu().clone() for k, v in ema_m.state_dict().items()}))
        # validation-style forward of the averaged model (train.py:149-153) against a freshly built model: a stale
        # packed conv weight in ema_m's engine would show here
        with torch.no_grad():
            y_ema = ema_m(xd)
            _, _, fresh = _tiny()
            fresh.load_state_dict(ema_m.state_dict(), strict=True)
            y_fresh = fresh.eval()(xd)
        assert torch.equal(y_ema, y_fresh), (it, float((y_ema - y_fresh).abs().max()))
    check_iterations(res, g, strict_tol=3e-5, per_tensor_cos_min=None, f64=f64_trace(g), verbose=True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ema_model_forward_follows_updates(dtype):
    """htrvt_ema_update writes the averaged weights through raw pointers (no _version bump): the averaged model's
    engine must drop its packed conv weights / bfloat16 casts, or validation runs on the first snapshot for ever"""
    from htrvt_amd.ema import ModelEma
    cfg, sd, m = _tiny(dtype)
    ema = ModelEma(m, 0.9999)
    x, _, _ = O.synthetic_batch(2, 64, 512, 80, cfg.num_patches, seed=5)
    xd = x.cuda()
    with torch.no_grad():
        y0 = ema.ema(xd).clone()
        for k, v in m.state_dict().items():
            if v.dtype != torch.int64 and k != "pos_embed":
                v.add_(torch.randn_like(v) * 0.05 * (v.abs().mean() + 1e-3))
        ema.update(m, num_updates=0)          # decay(0) = 0.1: the average moves 90 % of the way
        y1 = ema.ema(xd)
        _, _, fresh = _tiny(dtype)
        fresh.load_state_dict(ema.ema.state_dict(), strict=True)
        y_fresh = fresh.eval()(xd)
    assert float((y1 - y0).abs().max()) > 1e-2      # the update is visible at all
    assert torch.equal(y1, y_fresh), float((y1 - y_fresh).abs().max())


def test_ema_update_bit_exact_against_torch():
    """ema*d + (1-d)*model with each product rounded (float32) and the int64 counters through float math + truncation:
    the kernel must reproduce torch's elementwise result bit for bit"""
    from htrvt_amd.ema import ModelEma
    from htrvt_amd.model import HTR_VT
    torch.manual_seed(3)
    m = HTR_VT.MaskedAutoencoderViT(80, img_size=[64, 512], patch_size=(4, 64), embed_dim=64, depth=1, num_heads=2,
                                    mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6)).cuda()
    ema = ModelEma(m, 0.9999)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if v.dtype == torch.int64:
                v.fill_(37)
            else:
                v.add_(torch.randn_like(v) * 0.1)
    want = {}
    for n_upd in (0.0, 7.5, 1e6):
        d = min(0.9999, (1 + n_upd) / (10 + n_upd))
        msd = m.state_dict()
        for k, e in ema.ema.state_dict().items():
            want[k] = (e * d + (1.0 - d) * msd[k]).to(e.dtype) if e.dtype != torch.int64 else (e * d + (1.0 - d) * msd[k]).to(torch.int64)
        ema.update(m, num_updates=n_upd)
        for k, e in ema.ema.state_dict().items():
            assert torch.equal(e, want[k]), (n_upd, k)


def test_greedy_decode_matches_reference_and_oracle(golden_dir):
    import htrvt_amd
    from htrvt_amd.ctc import greedy_decode
    g = np.load(os.path.join(golden_dir, "greedy_decode.npz"))
    idx, lens = greedy_decode(torch.from_numpy(g["logits"]).cuda())
    idx, lens = idx.cpu().numpy(), lens.cpu().numpy()
    assert np.array_equal(lens, g["lens"])
    assert np.array_equal(np.concatenate([idx[b, :lens[b]] for b in range(len(lens))]), g["flat"])
    # a larger random case against the oracle, T = 512, nb_cls = 90 with the converter's index cut-off
    rng = np.random.default_rng(9)
    logits = rng.standard_normal((16, 512, 90)).astype(np.float32)
    logits[:, :, 0] += 2.0
    idx, lens = greedy_decode(torch.from_numpy(logits).cuda())
    want = O.greedy_decode(logits)
    for b in range(16):
        assert idx[b, :int(lens[b])].cpu().tolist() == want[b]
    idx2, lens2 = greedy_decode(torch.from_numpy(logits).cuda(), ncharacter=60)   # t[i] < len(character), utils.py:80
    am = logits.argmax(2)
    for b in range(16):
        seq = [int(v) for i, v in enumerate(am[b]) if v != 0 and not (i > 0 and am[b][i - 1] == v) and v < 60]
        assert idx2[b, :int(lens2[b])].cpu().tolist() == seq
