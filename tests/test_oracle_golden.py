"""Pins oracle/htrvt_oracle.py against vectors produced by the reference itself
(tools/make_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import htrvt_oracle as O

TOL = 2e-5   # fp32 op-order noise between two CPU evaluations of the same math


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_tiny_model_forward_eval_and_train(golden_dir):
    g = _load(golden_dir, "tiny_model.npz")
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    sd = O.init_state_dict(cfg, seed=7, randomize_affine=True)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        y = O.forward(sd, cfg, x, train=False)
        yt = O.forward(sd, cfg, x, keep_mask=torch.from_numpy(g["keep_mask"]), train=True)
    assert np.abs(y.numpy() - g["logits_eval"]).max() < TOL
    assert np.abs(yt.numpy() - g["logits_train"]).max() < TOL


def test_tiny_model_loss_grads_and_bn_stats(golden_dir):
    g = _load(golden_dir, "tiny_model.npz")
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    sd = O.init_state_dict(cfg, seed=7, randomize_affine=True)
    loss, logits, grads, stats = O.loss_and_grads(sd, cfg, torch.from_numpy(g["x"]), g["targets"], g["lengths"],
                                                  keep_mask=torch.from_numpy(g["keep_mask"]), train=True)
    assert abs(loss - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
    n = 0
    for k in g.files:
        if not k.startswith("grad."):
            continue
        ref = g[k]
        got = grads[k[5:]].numpy()
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(got - ref).max() / scale < 2e-3, k
        n += 1
    assert n == 77
    # running stats after one train step: momentum 0.1, unbiased running var
    for prefix, (mean, var_unb) in stats.items():
        rm = 0.9 * sd[prefix + ".running_mean"] + 0.1 * mean
        rv = 0.9 * sd[prefix + ".running_var"] + 0.1 * var_unb
        assert np.abs(rm.numpy() - g["post." + prefix + ".running_mean"]).max() < 1e-4
        assert np.abs(rv.numpy() - g["post." + prefix + ".running_var"]).max() < 1e-3


@pytest.mark.parametrize("tag", ["cfg1_d256", "ref_d768", "d512_12L"])
def test_real_width_models(golden_dir, tag):
    g = _load(golden_dir, tag + ".npz")
    nb, H, W, D, depth, heads, B, wseed, xseed = [int(v) for v in g["meta"]]
    cfg = O.Config(nb, (H, W), embed_dim=D, depth=depth, num_heads=heads)
    sd = O.init_state_dict(cfg, seed=wseed, randomize_affine=True)
    x, _, _ = O.synthetic_batch(B, H, W, nb, cfg.num_patches, seed=xseed)
    with torch.no_grad():
        y = O.forward(sd, cfg, x, train=False)
        yt = O.forward(sd, cfg, x, keep_mask=torch.from_numpy(g["keep_mask"]), train=True)
    assert np.abs(y.numpy() - g["logits_eval"]).max() < 5e-5
    assert np.abs(yt.numpy() - g["logits_train"]).max() < 5e-5


def test_state_dict_contract(golden_dir):
    g = _load(golden_dir, "create_model_init.npz")
    cfg = O.Config(80, (64, 512))
    spec = O.state_dict_spec(cfg)
    assert [s[0] for s in spec] == [str(k) for k in g["keys"]]
    assert [str(tuple(s[1])) for s in spec] == [str(s) for s in g["shapes"]]
    assert len(spec) == 150


def test_pos_embed_exact(golden_dir):
    g = _load(golden_dir, "host_helpers.npz")
    pe = O.pos_embed_table(768, (16, 8)).astype(np.float32)
    assert np.array_equal(pe, g["pos_embed_768_128"][0])


def test_lr_schedule(golden_dir):
    g = _load(golden_dir, "host_helpers.npz")
    for it, lr in zip(g["lr_iters"], g["lr_values"]):
        assert abs(O.lr_cos(int(it), 1000, 100000, 1e-3) - float(lr)) < 1e-15


@pytest.mark.parametrize("name", ["ragged", "repeats", "infeasible", "t256"])
def test_ctc_known_answers(golden_dir, name):
    g = _load(golden_dir, "ctc_cases.npz")
    nll, mean, grad = O.ctc_loss(g[name + ".logits"], g[name + ".targets"], g[name + ".lengths"])
    ref = g[name + ".nll"]
    # the golden is ATen's fp32 CTC (abs error ~ eps32 * |nll| ~ 3e-4 at nll ~ 1e3); the oracle is
    # fp64 and agrees with ATen's own fp64 CTC to 1e-12 (checked in tools/)
    assert np.abs(nll - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    assert np.abs(grad - g[name + ".grad"]).max() < 1e-3 * np.abs(g[name + ".grad"]).max()
    # integer label expansion is exact
    ext = O.expand_labels(g[name + ".targets"], g[name + ".lengths"])
    off = 0
    for e, L in zip(ext, g[name + ".lengths"]):
        assert e.dtype == np.int64 and e.shape[0] == 2 * L + 1
        assert np.array_equal(e[1::2], g[name + ".targets"][off:off + L]) and not e[0::2].any()
        off += L


def test_span_mask_matches_reference_rng(golden_dir):
    g = _load(golden_dir, "tiny_model.npz")
    torch.manual_seed(11)
    m = O.span_mask(128, 0.4, 8)
    assert np.array_equal(m.numpy(), g["keep_mask"])
    assert int((m == 0).sum()) <= 6 * 8 and int((m == 0).sum()) >= 8
