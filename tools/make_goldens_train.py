#!/usr/bin/env python3
"""Generate tests/golden/train_iter.npz and greedy_decode.npz by running the REFERENCE's training iteration
(SAM(AdamW) two-pass step + ModelEma, model_v1/train.py:119-128, utils/sam.py, utils/utils.py:128-173) and its
greedy CTC decode (valid.py:40-42, utils/utils.py:72-86) on CPU.  Dev container only (needs /root/reference); the
fixtures are data.  Same in-memory timm stub as tools/make_goldens.py.

    python tools/make_goldens_train.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = "/root/reference/model_v1"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    from make_goldens import _install_timm_stub
    _install_timm_stub()
    sys.path.insert(0, REF)
    from model import HTR_VT as REF_HTR_VT          # noqa: E402  (the reference)
    from utils import utils as ref_utils            # noqa: E402
    from utils import sam as ref_sam                # noqa: E402
    from oracle import htrvt_oracle as O            # noqa: E402  (only for the weight / batch generators)
    from functools import partial

    torch.set_num_threads(8)
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    sd = O.init_state_dict(cfg, seed=7, randomize_affine=True)
    m = REF_HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch, embed_dim=cfg.D,
                                        depth=cfg.depth, num_heads=cfg.heads, mlp_ratio=4,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6))
    m.load_state_dict(sd, strict=True)
    m.train()
    x, targets, lengths = O.synthetic_batch(4, 64, 512, 80, cfg.num_patches, seed=3)
    ema = ref_utils.ModelEma(m, 0.9999)
    opt = ref_sam.SAM(m.parameters(), torch.optim.AdamW, lr=1e-7, betas=(0.9, 0.99), weight_decay=0.5)
    crit = torch.nn.CTCLoss(reduction="none", zero_infinity=True)
    LR = 1e-3
    out = {"lr": np.float64(LR), "rho": np.float64(0.05)}

    def loss_of(seed):
        torch.manual_seed(seed)                     # the span mask comes from the CPU generator
        y = m(x, 0.4, 8, use_masking=True).float()
        lp = y.permute(1, 0, 2).log_softmax(2)
        return crit(lp, torch.from_numpy(targets), torch.IntTensor([lp.shape[0]] * 4), torch.from_numpy(lengths)).mean()

    for it in range(2):
        for g in opt.param_groups:                  # utils.update_lr_cos writes the lr the same way
            g["lr"] = LR
        opt.zero_grad()
        loss = loss_of(100 + 2 * it)
        loss.backward()
        opt.first_step(zero_grad=True)
        loss_of(101 + 2 * it).backward()
        opt.second_step(zero_grad=True)
        m.zero_grad()
        ema.update(m, num_updates=it / 2)
        out[f"it{it}.loss"] = np.float32(loss.item())
        for k, v in m.state_dict().items():
            out[f"it{it}.model.{k}"] = v.detach().numpy().copy()
        for k, v in ema.ema.state_dict().items():
            out[f"it{it}.ema.{k}"] = v.detach().numpy().copy()
        print("iteration", it, "loss", float(loss))
    np.savez_compressed(os.path.join(OUT, "train_iter.npz"), **out)

    # ---- greedy decode: valid.py:40-42 + CTCLabelConverter.decode ---------------------------------------
    chars = [chr(33 + i) for i in range(79)]        # 79 symbols -> indices 1..79, 0 = blank
    conv = ref_utils.CTCLabelConverter(chars)
    rng = np.random.default_rng(4)
    B, T, C = 6, 128, 80
    logits = rng.standard_normal((B, T, C)).astype(np.float32)
    logits[:, :, 0] += 1.5                          # plenty of blanks
    logits[0, 10:20, 5] += 9.0                      # a long repeat
    logits[1, :, 0] += 50.0                         # all blank -> empty string
    logits[2, ::2, 7] += 9.0                        # same symbol separated by other frames
    lp = torch.from_numpy(logits).permute(1, 0, 2).log_softmax(2)
    _, idx = lp.max(2)
    idx = idx.transpose(1, 0).contiguous().view(-1)
    strs = conv.decode(idx.data, torch.IntTensor([T] * B).data)
    seqs = [np.array([conv.dict[c] for c in s], dtype=np.int32) for s in strs]
    np.savez_compressed(os.path.join(OUT, "greedy_decode.npz"), logits=logits,
                        flat=np.concatenate(seqs + [np.zeros(0, np.int32)]), lens=np.array([len(s) for s in seqs], np.int32))
    print("decode lens", [len(s) for s in seqs])


if __name__ == "__main__":
    main()
