#!/usr/bin/env python3
"""How long does the HOST take to enqueue one training step (no synchronisation inside the loop), against the device time
of the step?  If the two are close the step is launch-bound and the GPU idles between kernels.
    python tools/host_time.py [--steps 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import htrvt_amd  # noqa: E402,F401
from htrvt_amd.model import HTR_VT  # noqa: E402
from htrvt_amd.trainer import Trainer  # noqa: E402
from oracle import htrvt_oracle as O  # noqa: E402  (synthetic batch only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    torch.manual_seed(123)
    m = HTR_VT.create_model(nb_cls=80, img_size=[64, 1024], compute_dtype=torch.bfloat16).cuda().train()
    N = m.patch_embed_num if hasattr(m, "patch_embed_num") else 256
    x, tg, tl = O.synthetic_batch(a.batch, 64, 1024, 80, 256, seed=0)
    x = x.cuda()
    tr = Trainer(m, max_lr=1e-3, weight_decay=0.5)
    keep = m.generate_span_mask(256, 0.4, 8)
    for _ in range(3):
        tr.step(x, tg, tl, keep_mask=keep)
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        h0 = time.perf_counter()
        tr.step(x, tg, tl, keep_mask=keep)
        host.append(time.perf_counter() - h0)
    enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"{a.steps} steps: host enqueue {enq / a.steps * 1e3:.2f} ms/step (min {min(host) * 1e3:.2f}, max {max(host) * 1e3:.2f}), "
          f"device-complete {total / a.steps * 1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
