#!/usr/bin/env python3
"""Error ladder of the bf16 throughput path at the headline shape (d768 / 4L / 6h, 64x1024, B = 128): where does the
logit error enter, and why is train mode (batch-statistics BatchNorm, span mask) several times worse than eval mode?

Two ladders, stage by stage (after conv1 + pool, each BasicBlock's conv outputs and output, tokens, each encoder block,
logits), each in eval and in train mode:

  gpu   the bf16 engine against the float32 engine (which is within 3e-5 of the oracle everywhere,
        tests/test_full_shape_gpu.py) -- both run with save=True, the saved activations are compared;
  cpu   the ORACLE under torch.autocast(bfloat16) against the oracle in float32 on a subset of the batch: what the
        reference itself loses to bf16 operands (BASELINE.md section 2 quotes 2.5e-2 for its eval forward).

    python tools/bf16_ladder.py --out profiles/r04_bf16_ladder.md
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def metrics(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    d = a - b
    return float(d.abs().max()), float(d.norm() / (b.norm() + 1e-30)), float(a @ b / (a.norm() * b.norm() + 1e-30))


def gpu_ladder(cfg, sd, x, keep, train, fold_eval=False):
    from functools import partial
    from htrvt_amd.model import HTR_VT
    acts = {}
    for dtype in (torch.float32, torch.bfloat16):
        m = HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch, embed_dim=cfg.D,
                                        depth=cfg.depth, num_heads=cfg.heads, mlp_ratio=4,
                                        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), compute_dtype=dtype)
        m.load_state_dict(sd, strict=True)
        m = m.cuda()
        eng = m._engine(torch.device("cuda"))
        P = dict(m.state_dict(keep_vars=True))
        with torch.no_grad():
            y = eng.forward(P, x, keep_mask=keep if train else None, train=train, save=True)
        sv, eng.saved = eng.saved, None
        st = [("conv1+bn+relu+pool", sv["stem_blocks"][0]["x"])]
        for blk in sv["stem_blocks"]:
            p = blk["p"].replace("patch_embed.", "")
            st += [(p + " conv1 (raw)", blk["ca"]), (p + " relu(bn1)", blk["a1"]), (p + " conv2 (raw)", blk["cb"]), (p + " out", blk["out"])]
        for e in sv["enc"]:
            st += [(e["p"] + " input", e["x0"]), (e["p"] + " attn out (O)", e["O"]), (e["p"] + " after attn", e["x1"]),
                   (e["p"] + " gelu(fc1)", e["h"])]
        st += [("encoder output", sv["x_last"]), ("final norm", sv["xn"]), ("logits", y)]
        acts[dtype] = [(n, t.float().cpu()) for n, t in st]
        del m, eng, sv, P
        torch.cuda.empty_cache()
    rows = []
    for (n, a16), (_, a32) in zip(acts[torch.bfloat16], acts[torch.float32]):
        rows.append((n,) + metrics(a16, a32))
    agree = float((acts[torch.bfloat16][-1][1].argmax(-1) == acts[torch.float32][-1][1].argmax(-1)).float().mean())
    return rows, agree


def cpu_ladder(cfg, sd, x, keep, train):
    from oracle import htrvt_oracle as O
    import torch.nn.functional as F
    taps_all = []
    for ac in (False, True):
        taps = {}
        orig_block = O._basic_block

        def rec_block(sd_, p, x_, stride, has_ds, train_, stats_out, _orig=orig_block, _t=taps):
            out = _orig(sd_, p, x_, stride, has_ds, train_, stats_out)
            _t[p.replace("patch_embed.", "") + " out"] = out.detach().float().permute(0, 2, 3, 1).clone()
            return out
        O._basic_block = rec_block
        try:
            with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16, enabled=ac):
                t2 = {}
                y = O.forward(sd, cfg, x, keep_mask=keep if train else None, train=train, taps=t2)
        finally:
            O._basic_block = orig_block
        taps.update({k: v.float() for k, v in t2.items()})
        taps["logits"] = y.float()
        taps_all.append(taps)
    f32, b16 = taps_all
    rows = [(n,) + metrics(b16[n], f32[n]) for n in f32]
    agree = float((b16["logits"].argmax(-1) == f32["logits"].argmax(-1)).float().mean())
    return rows, agree


def table(rows):
    out = ["| stage | max-abs | rel-L2 | cosine |", "|---|---:|---:|---:|"]
    for n, mx, l2, cs in rows:
        out.append(f"| {n} | {mx:.3e} | {l2:.3e} | {cs:.6f} |")
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-gpu", action="store_true")
    args = ap.parse_args()
    from oracle import htrvt_oracle as O
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    cfg = O.Config(80, (64, args.width), embed_dim=768, depth=4, num_heads=6)
    sd = O.init_state_dict(cfg, seed=123, randomize_affine=True)
    x, _, _ = O.synthetic_batch(args.batch, 64, args.width, 80, cfg.num_patches, seed=0)
    torch.manual_seed(7)
    keep = O.span_mask(cfg.num_patches, 0.4, 8)
    lines = [f"# bf16 error ladder, d768/4L/6h, 64x{args.width}, B = {args.batch} (tools/bf16_ladder.py)", ""]
    for train in (False, True):
        mode = "train (batch-statistics BatchNorm, span mask 0.4/8)" if train else "eval (running statistics)"
        if not args.no_gpu:
            rows, agree = gpu_ladder(cfg, sd, x.cuda(), keep, train)
            lines += [f"## GPU: bf16 engine vs float32 engine, {mode}", "", f"arg-max agreement of the logits: {agree:.4f}", "", table(rows), ""]
        rows, agree = cpu_ladder(cfg, sd, x[:args.cpu_batch], keep, train)
        lines += [f"## CPU: the oracle under torch.autocast(bfloat16) vs the oracle in float32, {mode}, first {args.cpu_batch} images",
                  "", f"arg-max agreement of the logits: {agree:.4f}", "", table(rows), ""]
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
