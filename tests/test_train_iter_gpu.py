"""The training iteration around the path on the GPU: Trainer.sam_step (SAM(AdamW) as flat HIP
launches), ModelEma.update (one multi-tensor launch) and the device greedy CTC decode, against goldens produced by the
reference (tools/make_goldens_train.py) -- same criteria as the CPU oracle test."""
import os
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import htrvt_oracle as O
from test_train_iter_cpu import check_iterations

pytestmark = pytest.mark.gpu


def test_sam_step_and_ema_match_reference_iterations(golden_dir):
    from htrvt_amd.ema import ModelEma
    from htrvt_amd.model import HTR_VT
    from htrvt_amd.trainer import Trainer
    g = np.load(os.path.join(golden_dir, "train_iter.npz"))
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    sd = O.init_state_dict(cfg, seed=7, randomize_affine=True)
    m = HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch, embed_dim=cfg.D,
                                    depth=cfg.depth, num_heads=cfg.heads, mlp_ratio=4,
                                    norm_layer=partial(nn.LayerNorm, eps=1e-6), compute_dtype=torch.float32)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    ema = ModelEma(m, 0.9999)
    tr = Trainer(m, max_lr=float(g["lr"]), betas=(0.9, 0.99), weight_decay=0.5)
    x, targets, lengths = O.synthetic_batch(4, 64, 512, 80, cfg.num_patches, seed=3)
    xd = x.cuda()
    res = []
    for it in range(2):
        masks = []
        for s in (100 + 2 * it, 101 + 2 * it):
            torch.manual_seed(s)
            masks.append(m.generate_span_mask(cfg.num_patches, 0.4, 8))
        loss = tr.sam_step(xd, targets, lengths, masks[0], masks[1], lr=float(g["lr"]), rho=float(g["rho"]))
        ema.update(m, num_updates=it / 2)
        res.append((float(loss), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
                    {k: v.detach().cpu().clone() for k, v in ema.ema.state_dict().items()}))
    check_iterations(res, g, strict_tol=3e-5, cos_min=0.9)   # conv1 (every ReLU / arg-max flip downstream of it) reaches 0.92


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
