#!/usr/bin/env python3
"""Generate tests/golden/line_prepare.npz: random grey scans of ragged sizes and what the reference's loader code
(data/dataset.py:104-135: PIL resize to height 64 keeping the aspect, img_as_float32, right pad with 1.0) makes of them,
computed with the installed Pillow.  skimage is not in this image; img_as_float32 of a uint8 array is value / 255
(restated), so the fixture stores the uint8 result (255 = 1.0).

    python tools/make_goldens_line.py"""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SIZES = [(37, 210), (64, 512), (64, 700), (128, 1500), (200, 900), (48, 1024), (91, 333), (64, 1024), (30, 50), (257, 4000)]


def np_thum(img, max_w, max_h):          # dataset.py:104-111, verbatim semantics
    x, y = np.shape(img)[:2]
    y = min(int(y * max_h / x), max_w)
    x = max_h
    return np.array(Image.fromarray(img).resize((y, x)))


def main():
    rng = np.random.default_rng(5)
    out = {"max_w": np.int32(1024), "max_h": np.int32(64), "n": np.int32(len(SIZES))}
    for i, (h, w) in enumerate(SIZES):
        # smooth background + strokes + noise: exercises negative bicubic lobes and the 0 / 255 clipping
        img = rng.integers(0, 256, (h, w)).astype(np.float64)
        img[h // 3: 2 * h // 3, ::7] = 0
        img[:, w // 2:] = np.clip(img[:, w // 2:] * 0.2 + 200, 0, 255)
        img = img.astype(np.uint8)
        r = np_thum(img, 1024, 64)
        padded = np.pad(r, ((0, 0), (0, 1024 - r.shape[1])), mode="constant", constant_values=255)
        out[f"src{i}"], out[f"dst{i}"] = img, padded
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "line_prepare.npz"), **out)
    print("wrote", len(SIZES), "cases")


if __name__ == "__main__":
    main()
