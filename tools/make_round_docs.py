#!/usr/bin/env python3
"""Formats the bench.py JSON lines / pytest -s logs a GPU call left under gpurun_out/<dir>/ into the round's committed
evidence files under profiles/ (small-batch table, BASELINE config table, parity log).
    python tools/make_round_docs.py gpurun_out/r03 r03"""
import json
import os
import sys


def load(path):
    try:
        with open(path) as f:
            lines = [l for l in f.read().strip().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    # ---- small batches: what one rank sees under --scaling strong at 2 / 4 / 8 GPUs
    rows = []
    for b in (128, 64, 32, 16):
        d = load(os.path.join(src, f"b{b}.json"))
        if d:
            rows.append((b, d))
    if rows:
        base = rows[0][1]["value"]
        with open(os.path.join(prof, f"{tag}_small_batch.md"), "w") as f:
            f.write(f"# Round {tag[1:]}: training step at the per-rank batch of a strong-scaling run (global batch 128)\n\n"
                    "`python bench.py --batch B --steps 10 --warmup 3` on ONE MI355X, bf16, 64x1024: the batch a rank holds when the\n"
                    "global batch of 128 is split over 1 / 2 / 4 / 8 GPUs (`bench.py --gpus N --scaling strong`).  The projected\n"
                    "N-GPU throughput is N x the single-GPU rate at B = 128 / N, i.e. it assumes the three bucketed gradient\n"
                    "all-reduces (214 MB float32 in total, 22 MB of it after the backward) stay hidden or small beside the step;\n"
                    "the measured curve is the driver's SCALE file.\n\n"
                    "| GPUs (strong) | batch per GPU | ms / step | images / s per GPU | projected images / s | projected speed-up | all MFMA launches TFLOP/s |\n"
                    "|---:|---:|---:|---:|---:|---:|---:|\n")
            for b, d in rows:
                n = 128 // b
                r = d.get("roofline") or {}
                f.write(f"| {n} | {b} | {d['ms_per_step']:.2f} | {d['value']:.0f} | {n * d['value']:.0f} | {n * d['value'] / base:.2f}x | {r.get('all_mfma_tflops')} |\n")
            f.write("\nWeak scaling (128 images per GPU, `bench.py --gpus N`, the default): every rank runs the B = 128 line.\n")
            f.write("\nJSON lines:\n\n```\n" + "\n".join(json.dumps(d) for _, d in rows) + "\n```\n")
    # ---- BASELINE configs
    cfgs = [("config 2: HTR-VT base eval forward + CTC, B=128, bf16", "cfg2_fwd"), ("headline: training step, B=128, bf16 (+ float32 parity path, CPU oracle)", "b128"),
            ("config 4: 64x2048 (N = 512 tokens), B=64, bf16 training step", "cfg4_w2048"),
            ("config 5: d512/12L/8h nb_cls 90, B=32 per GPU, bf16 training step", "cfg5_bf16"),
            ("config 5: same, float32 parity path", "cfg5_f32"), ("reference iteration: SAM(AdamW) 2x(fwd+bwd) + EMA, B=128, bf16", "sam")]
    got = [(t, load(os.path.join(src, n + ".json"))) for t, n in cfgs]
    if any(d for _, d in got):
        with open(os.path.join(prof, f"{tag}_configs.md"), "w") as f:
            f.write(f"# Round {tag[1:]}: bench.py lines for the BASELINE.json configurations (one MI355X)\n\n"
                    "| configuration | metric | ms / step | images / s | dtype |\n|---|---|---:|---:|---|\n")
            for t, d in got:
                if d:
                    f.write(f"| {t} | {d['metric']} | {d['ms_per_step']:.2f} | {d['value']:.0f} | {d['dtype']} |\n")
            f.write("\nCommands: `bench.py --forward-only`; `bench.py`; `bench.py --width 2048 --batch 64`; `bench.py --embed-dim 512 --depth 12 "
                    "--heads 8 --nb-cls 90 --batch 32 [--dtype f32]`; `bench.py --sam`.\n\nJSON lines:\n\n```\n")
            f.write("\n".join(json.dumps(d) for _, d in got if d) + "\n```\n")
    # ---- parity log
    plog = os.path.join(src, "parity.log")
    if os.path.exists(plog):
        keep = [l.rstrip() for l in open(plog) if any(k in l for k in ("max-abs", "cosine", "agreement", "passed", "failed", "loss", "decisions", "rel-L2"))]
        with open(os.path.join(prof, f"{tag}_parity.md"), "w") as f:
            f.write(f"# Round {tag[1:]}: parity numbers printed by the GPU tests\n\n"
                    "`python -m pytest tests/test_full_shape_gpu.py tests/test_model_gpu.py tests/test_train_iter_gpu.py -m gpu -s -q` on one MI355X "
                    "(the lines the tests print; gates are in the tests).\n\n```\n" + "\n".join(keep) + "\n```\n")


if __name__ == "__main__":
    main()
