#!/usr/bin/env python3
"""Time the fused CTC loss + gradient (csrc/ctc.hip) at the bench shape: B samples, T frames, C classes, synthetic targets
drawn as oracle.synthetic_batch does (20..89 labels).   python tools/bench_ctc.py [B T C]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import htrvt_amd  # noqa: E402
from htrvt_amd.ctc import ctc_forward_backward, stage_targets  # noqa: E402


def main():
    B, T, C = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (128, 256, 80)
    rng = np.random.default_rng(0)
    hi = max(3, min(90, T // 2))
    lengths = rng.integers(min(20, hi - 1), hi, size=B).astype(np.int32)
    targets = rng.integers(1, C, size=int(lengths.sum())).astype(np.int32)
    logits = torch.randn(B, T, C, device="cuda")
    staged = stage_targets(targets, lengths, logits.device)
    for want in (True, False):
        for _ in range(3):
            ctc_forward_backward(logits, targets, lengths, want_grad=want, staged=staged)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            ctc_forward_backward(logits, targets, lengths, want_grad=want, staged=staged)
        e1.record()
        torch.cuda.synchronize()
        print(f"B={B} T={T} C={C} max target {int(lengths.max())}: {'loss + gradient' if want else 'loss only'} "
              f"{e0.elapsed_time(e1) / n * 1e3:.1f} us per call (allocations included)")


if __name__ == "__main__":
    main()
