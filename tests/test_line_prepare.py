"""Input hand-off: resize to height 64 keeping the aspect, value / 255, right pad with 1.0
(reference data/dataset.py:104-135).  CPU: the oracle restatement of Pillow's 8-bit bicubic resampler against the
fixtures produced with Pillow itself (tools/make_goldens_line.py) and against the installed Pillow on fresh random
scans.  GPU: htrvt_line_prepare (through htrvt_amd.prepare_lines) bit for bit against both, and the model on the
prepared uint8 batch against the model on the float batch the reference loader would have built."""
import os
from functools import partial

import numpy as np
import pytest
import torch

from oracle import line_prepare_oracle as L


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "line_prepare.npz"))
    return [(g[f"src{i}"], g[f"dst{i}"]) for i in range(int(g["n"]))], int(g["max_w"]), int(g["max_h"])


def test_oracle_matches_pillow_fixtures(golden_dir):
    cases, max_w, max_h = _cases(golden_dir)
    for src, dst in cases:
        assert np.array_equal(L.prepare_line(src, max_w, max_h), dst), src.shape


def test_oracle_matches_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    for h, w in [(64, 64), (20, 300), (333, 2000), (64, 1500), (100, 100), (7, 9)]:
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
        ow = L.thumb_width(h, w, 64, 1024)
        want = np.array(Image.fromarray(img).resize((ow, 64)))
        assert np.array_equal(L.pil_resize_bicubic_u8(img, ow, 64), want), (h, w)


@pytest.mark.gpu
def test_kernel_matches_fixtures_and_oracle(golden_dir):
    import htrvt_amd
    cases, max_w, max_h = _cases(golden_dir)
    out = htrvt_amd.prepare_lines([s for s, _ in cases], max_w, max_h)
    assert out.shape == (len(cases), 1, max_h, max_w) and out.dtype == torch.uint8
    got = out.cpu().numpy()
    for i, (src, dst) in enumerate(cases):
        assert np.array_equal(got[i, 0], dst), (i, src.shape, int(np.abs(got[i, 0].astype(int) - dst).max()))
    # a fresh ragged batch at another model width, against the oracle
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (int(rng.integers(20, 200)), int(rng.integers(40, 1800)))).astype(np.uint8) for _ in range(24)]
    imgs += [rng.integers(0, 256, (64, 512)).astype(np.uint8), rng.integers(0, 256, (64, 10)).astype(np.uint8)]   # no resize at all
    got = htrvt_amd.prepare_lines(imgs, 512, 64).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(got[i, 0], L.prepare_line(im, 512, 64)), (i, im.shape)
    with pytest.raises(ValueError):
        htrvt_amd.prepare_lines([np.zeros((64 * 20, 64), np.uint8)], 512, 64)      # shrinks by more than the tap table covers


@pytest.mark.gpu
def test_c_abi_over_scale_image_is_blanked_not_overrun():
    """htrvt_line_prepare is a public C entry whose image table lives in device memory: an image that shrinks by more than
    htrvt_line_max_scale() (more taps than the LDS tables hold) must come out as an empty line (all 255), and its
    neighbours in the batch must be untouched -- prepare_lines() refuses such an image earlier, so go through the C ABI"""
    import ctypes as C
    from htrvt_amd._lib import check, lib
    from htrvt_amd.data import _LineImage
    from htrvt_amd.ops import ptr, stream
    rng = np.random.default_rng(5)
    H, W = 64, 256
    smax = lib.htrvt_line_max_scale()
    good = rng.integers(0, 256, (90, 300)).astype(np.uint8)
    tall = rng.integers(0, 256, (H * (smax + 3), 200)).astype(np.uint8)      # vertical scale > smax
    wide = rng.integers(0, 256, (64, 256 * (smax + 2))).astype(np.uint8)      # horizontal scale > smax (width' = W)
    arrs = [good, tall, wide, good[::-1].copy()]
    table = (_LineImage * len(arrs))()
    so = to = 0
    for i, a in enumerate(arrs):
        table[i] = _LineImage(so, to, a.shape[0], a.shape[1])
        so += a.size
        to += a.shape[0] * W
    src = torch.from_numpy(np.concatenate([a.reshape(-1) for a in arrs])).cuda()
    tab = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).cuda()
    tmp = torch.empty(to, dtype=torch.uint8, device="cuda")
    dst = torch.zeros(len(arrs), 1, H, W, dtype=torch.uint8, device="cuda")
    check(lib.htrvt_line_prepare(ptr(src), ptr(tab), ptr(tmp), ptr(dst), len(arrs), H, W, max(a.shape[0] for a in arrs), stream()),
          "line_prepare")
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    assert np.array_equal(got[0, 0], L.prepare_line(good, W, H))
    assert np.array_equal(got[3, 0], L.prepare_line(arrs[3], W, H))
    assert (got[1] == 255).all() and (got[2] == 255).all()


@pytest.mark.gpu
def test_model_on_prepared_batch_equals_reference_loader_path():
    """loader semantics end to end: the float batch the reference builds on the host (resize, / 255, pad 1.0) and the
    uint8 batch of prepare_lines give the same logits bit for bit"""
    import torch.nn as nn
    import htrvt_amd
    from htrvt_amd.model import HTR_VT
    from oracle import htrvt_oracle as O
    rng = np.random.default_rng(8)
    imgs = [rng.integers(0, 256, (int(rng.integers(30, 120)), int(rng.integers(200, 900)))).astype(np.uint8) for _ in range(4)]
    u8 = htrvt_amd.prepare_lines(imgs, 512, 64)
    ref = torch.from_numpy(np.stack([L.prepare_line(im, 512, 64) for im in imgs])[:, None].astype(np.float32) / np.float32(255.0))
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    m = HTR_VT.MaskedAutoencoderViT(80, img_size=[64, 512], patch_size=(4, 64), embed_dim=64, depth=2, num_heads=2, mlp_ratio=4,
                                    norm_layer=partial(nn.LayerNorm, eps=1e-6))
    m.load_state_dict(O.init_state_dict(cfg, seed=7, randomize_affine=True), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        y_u8, y_f = m(u8), m(ref.cuda())
    assert torch.equal(y_u8, y_f)
    y_ref = O.forward(O.init_state_dict(cfg, seed=7, randomize_affine=True), cfg, ref, train=False)
    assert (y_u8.cpu() - y_ref).abs().max() < 1e-3
