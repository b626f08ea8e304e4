#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Runs only in the dev container (needs /root/reference); the fixtures it writes
are data (inputs + expected outputs), never reference source.  The reference
imports `timm.models.vision_transformer.{Mlp,DropPath}` (timm==1.0.9, absent
here): an in-memory module restating timm's published Mlp
(fc1 -> act -> drop -> norm(Identity) -> fc2 -> drop) is injected for the import.

    python tools/make_goldens.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/model_v1"
OUT = os.path.join(ROOT, "tests", "golden")


def _install_timm_stub():
    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None,
                     act_layer=nn.GELU, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop)
            self.norm = nn.Identity()
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop2 = nn.Dropout(drop)

        def forward(self, x):
            return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            assert p == 0.0

        def forward(self, x):
            return x

    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.Mlp, vt.DropPath = Mlp, DropPath
    timm.models, models.vision_transformer = models, vt
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.vision_transformer": vt})


def main():
    _install_timm_stub()
    sys.path.insert(0, REF)
    from model import HTR_VT as REF_HTR_VT          # noqa: E402  (the reference)
    from utils import utils as ref_utils            # noqa: E402
    from oracle import htrvt_oracle as O            # noqa: E402  (only for the weight generator)
    from functools import partial

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    torch.manual_seed(0)

    def build(cfg, sd):
        m = REF_HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch,
                                            embed_dim=cfg.D, depth=cfg.depth, num_heads=cfg.heads,
                                            mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6))
        missing = m.load_state_dict(sd, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        return m

    def run_masked(m, x, ratio, span, seed):
        """train-mode forward with the reference's own CPU-RNG span mask; the
        mask it drew is re-derived from the same seed and recorded."""
        torch.manual_seed(seed)
        mask = O.span_mask(m.num_patches, ratio, span)      # consumes the RNG exactly like the reference
        torch.manual_seed(seed)
        y = m(x, ratio, span, use_masking=True)
        return y, mask

    # ---- (1) tiny full model: eval + train forward, CTC, every gradient ----------
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    sd = O.init_state_dict(cfg, seed=7, randomize_affine=True)
    m = build(cfg, sd)
    x, targets, lengths = O.synthetic_batch(4, 64, 512, 80, cfg.num_patches, seed=3)
    m.eval()
    with torch.no_grad():
        y_eval = m(x)
    m.train()
    y_train, mask = run_masked(m, x, 0.4, 8, seed=11)
    lp = y_train.float().permute(1, 0, 2).log_softmax(2)
    crit = torch.nn.CTCLoss(reduction="none", zero_infinity=True)
    per = crit(lp, torch.from_numpy(targets), torch.IntTensor([lp.shape[0]] * 4), torch.from_numpy(lengths))
    loss = per.mean()
    loss.backward()
    out = {"x": x.numpy(), "targets": targets, "lengths": lengths, "keep_mask": mask.numpy(),
           "logits_eval": y_eval.numpy(), "logits_train": y_train.detach().numpy(),
           "ctc_per_sample": per.detach().numpy(), "loss": np.float32(loss.item())}
    for k, v in m.state_dict().items():                 # BN running stats AFTER the train step
        if "running_" in k or "num_batches" in k:
            out["post." + k] = v.numpy()
    for k, p in m.named_parameters():
        if p.grad is not None:
            out["grad." + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_model.npz"), **out)
    print("tiny:", y_eval.shape, float(loss), len([k for k in out if k.startswith("grad.")]), "grads")

    # ---- (2) real-width models: weights regenerated from seed, only logits stored --
    for tag, kw, B, W in (("cfg1_d256", dict(embed_dim=256, depth=4, num_heads=4), 2, 512),
                          ("ref_d768", dict(embed_dim=768, depth=4, num_heads=6), 2, 512),
                          ("d512_12L", dict(embed_dim=512, depth=12, num_heads=8), 1, 512)):
        nb = 90 if tag == "d512_12L" else 80
        cfg = O.Config(nb, (64, W), **kw)
        sd = O.init_state_dict(cfg, seed=21, randomize_affine=True)
        m = build(cfg, sd)
        x, targets, lengths = O.synthetic_batch(B, 64, W, nb, cfg.num_patches, seed=5)
        m.eval()
        with torch.no_grad():
            y_eval = m(x)
        m.train()
        with torch.no_grad():
            y_train, mask = run_masked(m, x, 0.4, 8, seed=13)
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), logits_eval=y_eval.numpy(),
                            logits_train=y_train.numpy(), keep_mask=mask.numpy(),
                            meta=np.array([nb, 64, W, kw["embed_dim"], kw["depth"], kw["num_heads"], B, 21, 5]))
        print(tag, y_eval.shape, float(y_eval.abs().max()))

    # ---- (2b) create_model surface + reference init at seed 123 (checksums only) ---
    torch.manual_seed(123)
    m = REF_HTR_VT.create_model(nb_cls=80, img_size=[64, 512])
    keys = list(m.state_dict().keys())
    sums = np.array([float(v.double().sum()) for v in m.state_dict().values()])
    abss = np.array([float(v.double().abs().sum()) for v in m.state_dict().values()])
    shapes = [tuple(v.shape) for v in m.state_dict().values()]
    np.savez_compressed(os.path.join(OUT, "create_model_init.npz"), keys=np.array(keys),
                        sums=sums, abssums=abss, shapes=np.array([str(s) for s in shapes]),
                        nparams=np.int64(sum(p.numel() for p in m.parameters())))
    print("create_model:", len(keys), "tensors", sum(p.numel() for p in m.parameters()), "params")

    # ---- (3) CTC known answers from torch.nn.CTCLoss (what train.py calls) ---------
    rng = np.random.default_rng(99)
    cases = {}
    specs = [("ragged", 6, 128, 80, [5, 17, 1, 40, 0, 63]),        # incl. L=0
             ("repeats", 3, 64, 12, [10, 20, 31]),
             ("infeasible", 3, 16, 20, [16, 9, 12]),                # L>T or repeats needing >T
             ("t256", 2, 256, 80, [90, 33]),
             ("t512", 2, 512, 90, [120, 7])]
    for name, B, T, C, lens in specs:
        logits = rng.normal(0, 2.0, size=(B, T, C)).astype(np.float32)
        lens = np.array(lens, dtype=np.int32)
        if name == "repeats":
            tg = np.concatenate([np.repeat(rng.integers(1, C, size=(l + 1) // 2), 2)[:l] for l in lens]).astype(np.int32)
        elif name == "infeasible":
            tg = np.concatenate([np.full(l, 3) if i else rng.integers(1, C, size=l) for i, l in enumerate(lens)]).astype(np.int32)
        else:
            tg = rng.integers(1, C, size=int(lens.sum())).astype(np.int32)
        lg = torch.from_numpy(logits).requires_grad_(True)
        lp = lg.permute(1, 0, 2).log_softmax(2)
        per = crit(lp, torch.from_numpy(tg), torch.IntTensor([T] * B), torch.from_numpy(lens))
        per.mean().backward()
        cases[name + ".logits"] = logits
        cases[name + ".targets"] = tg
        cases[name + ".lengths"] = lens
        cases[name + ".nll"] = per.detach().numpy()
        cases[name + ".grad"] = lg.grad.numpy()
        print("ctc", name, per.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "ctc_cases.npz"), **cases)

    # ---- (4) host-side helpers -----------------------------------------------------
    class _Opt:
        param_groups = [{"lr": 0.0}]
    its = np.array([0, 1, 500, 999, 1000, 1001, 50000, 99999])
    lrs = np.array([ref_utils.update_lr_cos(int(i), 1000, 100000, 1e-3, _Opt())[1] for i in its])
    pe = m.pos_embed.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "host_helpers.npz"), lr_iters=its, lr_values=lrs,
                        pos_embed_768_128=pe)
    print("done ->", OUT)


if __name__ == "__main__":
    main()
