"""Inputs of the variant-attention fixtures (tests/golden/variants.npz), regenerated from seeds on both sides: the tool
that runs the reference (tools/make_goldens_variants.py) and the tests; only the reference's OUTPUTS are stored."""
import numpy as np

WINDOW_CASES = [  # tag, B, N, dim, heads, num_patches, window_size, shift_size
    ("full", 2, 128, 64, 2, 128, 0, 0),
    ("ws16", 2, 128, 64, 2, 128, 16, 0),
    ("ws16s8", 2, 128, 128, 4, 128, 16, 8),
    ("pad_ws16", 2, 120, 64, 2, 128, 16, 0),          # N not a multiple of the window: zero padding + key_padding_mask
    ("pad_ws16s8", 2, 120, 128, 4, 128, 16, 8),
    ("pad_ws48s5", 1, 100, 64, 2, 128, 48, 5),
]
SGM_CASES = [("d768", 2, 20, 32, 768), ("d64", 3, 5, 64, 64)]      # tag, B, L (queries), N (visual tokens), D


def window_inputs(case):
    tag, B, N, dim, heads, P, ws, shift = case
    r = np.random.default_rng(1000 + N + dim + ws + shift)
    s = 1.0 / np.sqrt(dim)
    return dict(x=r.standard_normal((B, N, dim)), gout=r.standard_normal((B, N, dim)),
                qkv_w=r.standard_normal((3 * dim, dim)) * s, qkv_b=r.standard_normal(3 * dim) * 0.1,
                proj_w=r.standard_normal((dim, dim)) * s, proj_b=r.standard_normal(dim) * 0.1,
                table=r.standard_normal((2 * P - 1, heads)) * 0.5)


def sgm_inputs(case):
    tag, B, L, N, D = case
    r = np.random.default_rng(2000 + L + D)
    return dict(Q=r.standard_normal((B, L, D)), F=r.standard_normal((B, N, D)), gout=r.standard_normal((B, L, D)),
                ln_w=1.0 + 0.2 * r.standard_normal(D), ln_b=0.2 * r.standard_normal(D))
