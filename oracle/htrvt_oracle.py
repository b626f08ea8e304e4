"""CPU oracle for the HTR-VT hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement of the reference's forward / loss
algorithm (plain torch functional ops on CPU tensors + numpy).  It is imported
only by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg, always as the checker, never as the product path.

Parity status: PINNED.  The reference (0xk0ry/HTR-VT) holds no tests, fixtures
or golden vectors for this path (``tests/*.py`` are 0 bytes), so the oracle is
pinned against outputs of the reference itself: ``tools/make_goldens.py``
imports ``/root/reference/model_v1`` on CPU and writes the input/output vectors
under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file
against them.

Reference lines each function follows (relative to /root/reference):
  whiten            model_v1/model/HTR_VT.py:134-136,224
  stem_forward      model_v1/model/resnet18.py:10-39,42-84
  pos_embed_table   model_v1/model/HTR_VT.py:86-131
  span_mask         model_v1/model/HTR_VT.py:202-210
  block_forward     model_v1/model/HTR_VT.py:27-39,80-83 (+ timm Mlp, timm==1.0.9)
  forward           model_v1/model/HTR_VT.py:222-241
  ctc_loss          model_v1/train.py:21-30 (torch.nn.CTCLoss, blank 0,
                    reduction='none', zero_infinity=True, then .mean())
  greedy_decode     model_v1/valid.py:40-42, model_v1/utils/utils.py:72-86
  lr_cos            model_v1/utils/utils.py:42-52
  init_state_dict   model_v1/model/HTR_VT.py:174-200 (init distributions only)
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-6        # create_model: partial(nn.LayerNorm, eps=1e-6)   HTR_VT.py:252
WHITEN_EPS = 1e-5    # param-free LayerNorm                             HTR_VT.py:136


# ----------------------------------------------------------------------------
# model description
# ----------------------------------------------------------------------------
class Config:
    """Shape description of one HTR-VT model (HTR_VT.py:143-172, 244-254)."""

    def __init__(self, nb_cls=80, img_size=(64, 512), embed_dim=768, depth=4,
                 num_heads=6, mlp_ratio=4.0, patch_size=(4, 64)):
        self.nb_cls = int(nb_cls)
        self.H, self.W = int(img_size[0]), int(img_size[1])
        self.D = int(embed_dim)
        self.depth = int(depth)
        self.heads = int(num_heads)
        self.hidden = int(embed_dim * mlp_ratio)
        self.patch = (int(patch_size[0]), int(patch_size[1]))
        self.grid = (self.H // self.patch[0], self.W // self.patch[1])
        self.num_patches = self.grid[0] * self.grid[1]

    @property
    def stem_channels(self):
        return (self.D // 4, self.D // 4, self.D // 2, self.D)


def state_dict_spec(cfg: Config):
    """Ordered (name, shape, kind) list == the reference module's state_dict
    (SURVEY.md 8(b)); kind in {'conv','bn_w','bn_b','bn_rm','bn_rv','bn_nbt',
    'lin_w','lin_b','ln_w','ln_b','mask_token','pos_embed'}."""
    D = cfg.D
    spec = [("mask_token", (1, 1, D), "mask_token"),
            ("pos_embed", (1, cfg.num_patches, D), "pos_embed")]

    def bn(prefix, c):
        return [(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
                (prefix + ".running_mean", (c,), "bn_rm"), (prefix + ".running_var", (c,), "bn_rv"),
                (prefix + ".num_batches_tracked", (), "bn_nbt")]

    c1 = D // 4
    spec.append(("patch_embed.conv1.weight", (c1, 1, 3, 3), "conv"))
    spec += bn("patch_embed.bn1", c1)
    inplanes = c1
    for li, planes in enumerate((D // 4, D // 2, D), start=1):
        for bi in range(2):
            p = f"patch_embed.layer{li}.{bi}"
            cin = inplanes if bi == 0 else planes
            spec.append((p + ".conv1.weight", (planes, cin, 3, 3), "conv"))
            spec += bn(p + ".bn1", planes)
            spec.append((p + ".conv2.weight", (planes, planes, 3, 3), "conv"))
            spec += bn(p + ".bn2", planes)
            if bi == 0:
                spec.append((p + ".downsample.0.weight", (planes, cin, 1, 1), "conv"))
                spec += bn(p + ".downsample.1", planes)
        inplanes = planes
    for i in range(cfg.depth):
        p = f"blocks.{i}"
        spec += [(p + ".norm1.weight", (D,), "ln_w"), (p + ".norm1.bias", (D,), "ln_b"),
                 (p + ".attn.qkv.weight", (3 * D, D), "lin_w"), (p + ".attn.qkv.bias", (3 * D,), "lin_b"),
                 (p + ".attn.proj.weight", (D, D), "lin_w"), (p + ".attn.proj.bias", (D,), "lin_b"),
                 (p + ".norm2.weight", (D,), "ln_w"), (p + ".norm2.bias", (D,), "ln_b"),
                 (p + ".mlp.fc1.weight", (cfg.hidden, D), "lin_w"), (p + ".mlp.fc1.bias", (cfg.hidden,), "lin_b"),
                 (p + ".mlp.fc2.weight", (D, cfg.hidden), "lin_w"), (p + ".mlp.fc2.bias", (D,), "lin_b")]
    spec += [("norm.weight", (D,), "ln_w"), ("norm.bias", (D,), "ln_b"),
             ("head.weight", (cfg.nb_cls, D), "lin_w"), ("head.bias", (cfg.nb_cls,), "lin_b")]
    return spec


def pos_embed_table(D: int, grid) -> np.ndarray:
    """2-D sin-cos table, float64 -> caller casts (HTR_VT.py:86-131).

    token t -> (h = t // gw, w = t % gw); first D/2 dims encode w, last D/2
    encode h (meshgrid is w-first, HTR_VT.py:94,106-109)."""
    gh, gw = int(grid[0]), int(grid[1])
    quarter = D // 4
    omega = 1.0 / 10000 ** (np.arange(quarter, dtype=np.float64) / (D / 4.0))
    t = np.arange(gh * gw)
    wpos = (t % gw).astype(np.float32).astype(np.float64)
    hpos = (t // gw).astype(np.float32).astype(np.float64)
    ow = wpos[:, None] * omega[None, :]
    oh = hpos[:, None] * omega[None, :]
    return np.concatenate([np.sin(ow), np.cos(ow), np.sin(oh), np.cos(oh)], axis=1)


def init_state_dict(cfg: Config, seed: int = 0, randomize_affine: bool = False):
    """Counter-free, torch-RNG-free weight generator with the reference's init
    DISTRIBUTIONS (HTR_VT.py:174-200, PyTorch default conv/BN init): used to
    regenerate identical weights on both sides of a golden without storing them.
    randomize_affine=True perturbs LN/BN affine + running stats + biases so that
    parity tests exercise every term (the reference init leaves them at 1/0)."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, shape, kind in state_dict_spec(cfg):
        if kind == "conv":
            fan_in = shape[1] * shape[2] * shape[3]
            b = 1.0 / math.sqrt(fan_in)
            v = rng.uniform(-b, b, size=shape)
        elif kind == "lin_w":
            b = math.sqrt(6.0 / (shape[0] + shape[1]))
            v = rng.uniform(-b, b, size=shape)
        elif kind in ("lin_b", "bn_b", "ln_b"):
            v = rng.uniform(-0.2, 0.2, size=shape) if randomize_affine else np.zeros(shape)
        elif kind in ("bn_w", "ln_w"):
            v = rng.uniform(0.6, 1.4, size=shape) if randomize_affine else np.ones(shape)
        elif kind == "bn_rm":
            v = rng.uniform(-0.3, 0.3, size=shape) if randomize_affine else np.zeros(shape)
        elif kind == "bn_rv":
            v = rng.uniform(0.5, 1.5, size=shape) if randomize_affine else np.ones(shape)
        elif kind == "bn_nbt":
            sd[name] = torch.tensor(0, dtype=torch.long)
            continue
        elif kind == "mask_token":
            v = rng.normal(0.0, 0.02, size=shape)
        elif kind == "pos_embed":
            v = pos_embed_table(cfg.D, cfg.grid)[None]
        else:
            raise KeyError(kind)
        sd[name] = torch.from_numpy(np.asarray(v, dtype=np.float64)).to(torch.float32)
    return sd


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------
def whiten(x):
    """Param-free LayerNorm over all non-batch dims, biased var, eps 1e-5
    (HTR_VT.py:134-136; called on the image :224 and on the logits :239)."""
    dims = tuple(range(1, x.dim()))
    mean = x.mean(dim=dims, keepdim=True)
    var = x.var(dim=dims, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + WHITEN_EPS)


def _bn(sd, prefix, x, train, stats_out):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if train:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            n = x.numel() // x.shape[1]
            stats_out[prefix] = (mean.detach().clone(), (var * n / max(n - 1, 1)).detach().clone())
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = w / torch.sqrt(var + BN_EPS)
    return x * scale[None, :, None, None] + (b - mean * scale)[None, :, None, None]


def _basic_block(sd, p, x, stride, has_ds, train, stats_out):
    """resnet18.py:23-39."""
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1)
    out = F.relu(_bn(sd, p + ".bn1", out, train, stats_out))
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, stride=1, padding=1)
    out = _bn(sd, p + ".bn2", out, train, stats_out)
    if has_ds:
        res = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride, padding=0)
        res = _bn(sd, p + ".downsample.1", res, train, stats_out)
    else:
        res = x
    return F.relu(out + res)


def stem_forward(sd, x, train=False, stats_out=None):
    """resnet18.py:73-84 on an already-whitened [B,1,H,W] image -> [B,D,1,W/4]."""
    p = "patch_embed"
    x = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=(2, 1), padding=1)
    x = F.relu(_bn(sd, p + ".bn1", x, train, stats_out))
    x = F.max_pool2d(x, kernel_size=3, stride=(2, 1), padding=1)
    for li, stride in ((1, (2, 1)), (2, (2, 2)), (3, (2, 2))):
        x = _basic_block(sd, f"{p}.layer{li}.0", x, stride, True, train, stats_out)
        x = _basic_block(sd, f"{p}.layer{li}.1", x, (1, 1), False, train, stats_out)
    return F.max_pool2d(x, kernel_size=3, stride=(2, 1), padding=1)


def span_mask(L: int, mask_ratio: float, max_span_length: int, generator=None):
    """HTR_VT.py:202-210: n = int(L*ratio)//span spans, start =
    torch.randint(L - span, (1,)) on the CPU generator, shared by the batch.
    Returns float32 [L] keep-mask (1 keep, 0 masked)."""
    mask = torch.ones(L)
    num_spans = int(L * mask_ratio) // max_span_length
    for _ in range(num_spans):
        idx = int(torch.randint(L - max_span_length, (1,), generator=generator))
        mask[idx:idx + max_span_length] = 0
    return mask


def layer_norm_rows(x, w, b, eps=LN_EPS):
    mean = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def attention(sd, p, x, heads):
    """HTR_VT.py:27-39."""
    B, N, D = x.shape
    hd = D // heads
    qkv = x @ sd[p + ".qkv.weight"].t() + sd[p + ".qkv.bias"]
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, D)
    return o @ sd[p + ".proj.weight"].t() + sd[p + ".proj.bias"]


def block_forward(sd, p, x, heads):
    """HTR_VT.py:80-83 with Identity ls*/drop_path* (init_values None, drop 0)."""
    x = x + attention(sd, p + ".attn", layer_norm_rows(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]), heads)
    h = layer_norm_rows(x, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    h = gelu_erf(h @ sd[p + ".mlp.fc1.weight"].t() + sd[p + ".mlp.fc1.bias"])
    return x + (h @ sd[p + ".mlp.fc2.weight"].t() + sd[p + ".mlp.fc2.bias"])


def forward(sd, cfg: Config, x, keep_mask=None, train=False, stats_out=None, taps=None):
    """HTR_VT.py:222-241.  x: [B,1,H,W]; keep_mask: None or float [N] (1 keep,
    0 -> mask_token), i.e. the recorded output of span_mask; train selects BN
    batch statistics.  Returns logits [B,N,nb_cls]."""
    x = whiten(x)
    f = stem_forward(sd, x, train=train, stats_out=stats_out)
    b, c = f.shape[0], f.shape[1]
    tok = f.reshape(b, c, -1).permute(0, 2, 1)
    if taps is not None:
        taps["tokens"] = tok.detach().clone()
    if keep_mask is not None:
        m = keep_mask.to(tok.dtype).reshape(1, -1, 1)
        tok = tok * m + (1 - m) * sd["mask_token"]
    tok = tok + sd["pos_embed"]
    for i in range(cfg.depth):
        tok = block_forward(sd, f"blocks.{i}", tok, cfg.heads)
        if taps is not None:
            taps[f"block{i}"] = tok.detach().clone()
    tok = layer_norm_rows(tok, sd["norm.weight"], sd["norm.bias"])
    logits = tok @ sd["head.weight"].t() + sd["head.bias"]
    return whiten(logits)


# ----------------------------------------------------------------------------
# CTC (numpy, log-space)
# ----------------------------------------------------------------------------
def _lse(*xs):
    m = max(xs)
    if m == -np.inf:
        return -np.inf
    return m + math.log(sum(math.exp(v - m) for v in xs))


def ctc_loss(logits: np.ndarray, targets: np.ndarray, target_lengths: np.ndarray,
             want_grad: bool = True):
    """Restates compute_loss (train.py:21-30): log_softmax over classes,
    CTC(blank=0, reduction='none', zero_infinity=True), mean over the batch
    (NOT length-normalised), input length = N for every sample.

    logits [B,N,C] float; targets 1-D int (concatenated); target_lengths [B] int.
    Returns (nll [B] float64, mean loss, dloss/dlogits [B,N,C] float64)."""
    logits = np.asarray(logits, dtype=np.float64)
    B, T, C = logits.shape
    mx = logits.max(axis=2, keepdims=True)
    lp = logits - mx - np.log(np.exp(logits - mx).sum(axis=2, keepdims=True))
    nll = np.zeros(B)
    grad = np.zeros_like(logits)
    off = 0
    for b in range(B):
        L = int(target_lengths[b])
        lab = np.asarray(targets[off:off + L], dtype=np.int64)
        off += L
        S = 2 * L + 1
        ext = np.zeros(S, dtype=np.int64)
        ext[1::2] = lab
        NEG = -np.inf
        alpha = np.full((T, S), NEG)
        alpha[0, 0] = lp[b, 0, 0]
        if S > 1:
            alpha[0, 1] = lp[b, 0, ext[1]]
        for t in range(1, T):
            for s in range(S):
                a = alpha[t - 1, s]
                a1 = alpha[t - 1, s - 1] if s >= 1 else NEG
                a2 = alpha[t - 1, s - 2] if (s >= 2 and ext[s] != 0 and ext[s] != ext[s - 2]) else NEG
                alpha[t, s] = _lse(a, a1, a2) + lp[b, t, ext[s]]
        ll = _lse(alpha[T - 1, S - 1], alpha[T - 1, S - 2] if S > 1 else NEG)
        if ll == NEG:           # infeasible: zero_infinity -> loss 0, grad 0
            nll[b] = 0.0
            continue
        nll[b] = -ll
        if not want_grad:
            continue
        beta = np.full((T, S), NEG)
        beta[T - 1, S - 1] = lp[b, T - 1, ext[S - 1]]
        if S > 1:
            beta[T - 1, S - 2] = lp[b, T - 1, ext[S - 2]]
        for t in range(T - 2, -1, -1):
            for s in range(S):
                v = beta[t + 1, s]
                v1 = beta[t + 1, s + 1] if s + 1 < S else NEG
                v2 = beta[t + 1, s + 2] if (s + 2 < S and ext[s + 2] != 0 and ext[s + 2] != ext[s]) else NEG
                beta[t, s] = _lse(v, v1, v2) + lp[b, t, ext[s]]
        occ = np.zeros((T, C))
        for s in range(S):
            ab = alpha[:, s] + beta[:, s]
            ok = ab > NEG
            occ[ok, ext[s]] += np.exp(ab[ok] - lp[b, ok, ext[s]] - ll)
        grad[b] = (np.exp(lp[b]) - occ) / B
    return nll, float(nll.mean()), grad


def expand_labels(targets, target_lengths):
    """Integer label expansion l' = [0,l1,0,l2,...,0] per sample (bit-exact
    part of the CTC path).  Returns list of int64 arrays."""
    out, off = [], 0
    for L in np.asarray(target_lengths).tolist():
        ext = np.zeros(2 * L + 1, dtype=np.int64)
        ext[1::2] = np.asarray(targets[off:off + L], dtype=np.int64)
        out.append(ext)
        off += L
    return out


def greedy_decode(logits: np.ndarray):
    """valid.py:40-42 + utils.py:72-86: argmax over classes, collapse repeats,
    drop blank 0.  Returns list of int lists (class ids)."""
    idx = np.asarray(logits).argmax(axis=2)
    res = []
    for row in idx:
        seq, prev = [], -1
        for v in row.tolist():
            if v != 0 and v != prev:
                seq.append(v)
            prev = v
        res.append(seq)
    return res


def lr_cos(nb_iter, warm_up_iter, total_iter, max_lr, min_lr=1e-7):
    """utils.py:42-52 (note: cosine phase uses nb_iter, not nb_iter-warm)."""
    if nb_iter < warm_up_iter:
        return max_lr * (nb_iter + 1) / (warm_up_iter + 1)
    return min_lr + (max_lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * nb_iter / (total_iter - warm_up_iter)))


# ----------------------------------------------------------------------------
# whole-step helpers used by tests / bench cpu_baseline
# ----------------------------------------------------------------------------
def loss_and_grads(sd, cfg: Config, x, targets, target_lengths, keep_mask=None, train=True, dtype=torch.float32):
    """fwd + CTC + bwd on CPU through torch autograd over this restatement
    (train.py:21-30,123).  dtype=torch.float64 evaluates the same graph in double
    (the reference is float32; the float64 run is the rounding-free yardstick).
    Returns (loss, logits, grads dict, bn batch stats)."""
    params = OrderedDict()
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            v = v.detach().to(dtype)
            if not k.endswith(("running_mean", "running_var")) and k != "pos_embed":
                v = v.clone().requires_grad_(True)
        params[k] = v
    stats = {}
    logits = forward(params, cfg, x.to(dtype), keep_mask=keep_mask, train=train, stats_out=stats)
    lp = logits.to(dtype).permute(1, 0, 2).log_softmax(2)
    T, B = lp.shape[0], lp.shape[1]
    loss = F.ctc_loss(lp, torch.as_tensor(targets, dtype=torch.int32),
                      torch.full((B,), T, dtype=torch.int32),
                      torch.as_tensor(target_lengths, dtype=torch.int32),
                      blank=0, reduction="none", zero_infinity=True).mean()
    loss.backward()
    grads = OrderedDict((k, v.grad.detach()) for k, v in params.items()
                        if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None)
    return float(loss.detach()), logits.detach(), grads, stats


def synthetic_batch(B, H, W, nb_cls, N, seed=0):
    """SURVEY.md 8(d) synthetic inputs: uniform [0,1) images, CTC-feasible
    random targets."""
    g = torch.Generator().manual_seed(1234 + seed)
    x = torch.rand(B, 1, H, W, generator=g)
    rng = np.random.default_rng(seed)
    hi = max(3, min(90, N // 2))
    lo = min(20, hi - 1)
    lengths = rng.integers(lo, hi, size=B).astype(np.int32)
    targets = rng.integers(1, nb_cls, size=int(lengths.sum())).astype(np.int32)
    return x, targets, lengths


# ----------------------------------------------------------------------------
# The training iteration around the path: SAM(AdamW) + EMA
# ----------------------------------------------------------------------------
def adamw_update(p, g, m, v, step, lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.5):
    """torch.optim.AdamW single-tensor step (train.py:94 hyper-parameters), in place on float32 tensors."""
    b1, b2 = betas
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def sam_adamw_iteration(sd, cfg: Config, x, targets, target_lengths, keep1, keep2, opt_state, lr, rho=0.05,
                        weight_decay=0.5, dtype=torch.float32):
    """One reference iteration (train.py:119-126 with utils/sam.py:15-38, adaptive=False) on a state_dict, in place:
    gradients at w -> w + rho g / (|g| + 1e-12) -> gradients there -> restore w -> AdamW with the second gradients.
    opt_state: {'step': int, 'm': {name: tensor}, 'v': {...}} (created empty on first use).
    BatchNorm running statistics advance twice, exactly as two train-mode forwards do.  Returns the first-pass loss.
    dtype=torch.float64 (with a float64 state_dict) evaluates the same iteration in double: the rounding-free yardstick
    for how far two float32 runs may legitimately drift apart (ReLU / arg-max flips under Adam's normalisation)."""
    def grads_at(keep):
        loss, _, grads, stats = loss_and_grads(sd, cfg, x, targets, target_lengths, keep, train=True, dtype=dtype)
        for k, (mean, var_unb) in stats.items():            # running-stat update of a train-mode forward
            sd[k + ".running_mean"].mul_(0.9).add_(mean, alpha=0.1)
            sd[k + ".running_var"].mul_(0.9).add_(var_unb, alpha=0.1)
            sd[k + ".num_batches_tracked"] += 1
        return loss, grads

    loss, g1 = grads_at(keep1)
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in g1.values())).to(dtype)
    scale = rho / (norm + 1e-12)
    old = {k: sd[k].clone() for k in g1}
    for k, g in g1.items():
        sd[k].add_(g * scale)
    _, g2 = grads_at(keep2)
    opt_state.setdefault("step", 0)
    opt_state["step"] += 1
    for k, g in g2.items():
        sd[k].copy_(old[k])
        m = opt_state.setdefault("m", {}).setdefault(k, torch.zeros_like(g))
        v = opt_state.setdefault("v", {}).setdefault(k, torch.zeros_like(g))
        adamw_update(sd[k], g, m, v, opt_state["step"], lr, weight_decay=weight_decay)
    return loss


def ema_update(ema_sd, model_sd, num_updates, decay=0.9999):
    """utils/utils.py:158-173: every state_dict entry, int64 counters included (float math, truncating copy_)."""
    d = min(decay, (1 + num_updates) / (10 + num_updates)) if num_updates >= 0 else decay
    for k, e in ema_sd.items():
        e.copy_(e * d + (1.0 - d) * model_sd[k])
