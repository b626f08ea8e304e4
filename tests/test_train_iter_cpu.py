"""Oracle restatement of the reference's training iteration (SAM(AdamW) two-pass step + ModelEma) and greedy CTC
decode against goldens produced by the reference itself (tools/make_goldens_train.py)."""
import math
import os

import numpy as np
import torch

from oracle import htrvt_oracle as O


def close_fraction(a, b, atol):
    return float(((a - b).abs() <= atol).double().mean())


_F64 = {}


def f64_trace(g):
    """parameters after iterations 0 and 1 with the whole iteration evaluated in float64 (the rounding-free yardstick)"""
    if "tr" not in _F64:
        _F64["tr"] = [sd for _, sd, _ in run_oracle_iterations(g, dtype=torch.float64)]
    return _F64["tr"]


def run_oracle_iterations(g, dtype=torch.float32):
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    sd = O.init_state_dict(cfg, seed=7, randomize_affine=True)
    if dtype != torch.float32:
        sd = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    ema_sd = {k: v.clone() for k, v in sd.items()}
    x, targets, lengths = O.synthetic_batch(4, 64, 512, 80, cfg.num_patches, seed=3)
    state, out = {}, []
    for it in range(2):
        masks = []
        for s in (100 + 2 * it, 101 + 2 * it):
            torch.manual_seed(s)
            masks.append(O.span_mask(cfg.num_patches, 0.4, 8))
        loss = O.sam_adamw_iteration(sd, cfg, x, targets, lengths, masks[0], masks[1], state, lr=float(g["lr"]), rho=float(g["rho"]),
                                     dtype=dtype)
        O.ema_update(ema_sd, sd, num_updates=it / 2)
        out.append((loss, {k: v.clone() for k, v in sd.items()}, {k: v.clone() for k, v in ema_sd.items()}))
    return out


def _key_bias_free(k, v):
    """the key bias of an attention layer has an identically zero gradient (softmax is invariant to it), so AdamW's
    sign-like first steps move it by +-lr on rounding noise alone: compare the query / value thirds only"""
    if k.endswith("attn.qkv.bias"):
        D = v.numel() // 3
        return torch.cat([v[:D], v[2 * D:]])
    return v


def _update_cosines(sd1, sd0, ref1, ref0, min_numel=64):
    """per-tensor and global cosine between two parameter updates (sd1 - sd0) and (ref1 - ref0)"""
    per, num, d1, d2 = {}, 0.0, 0.0, 0.0
    for k, v in sd1.items():
        if not v.dtype.is_floating_point or "running_" in k or k == "pos_embed":
            continue
        du = _key_bias_free(k, (v.double() - sd0[k].double()).flatten())
        dr = _key_bias_free(k, (ref1[k].double() - ref0[k].double()).flatten())
        if du.numel() >= min_numel:
            per[k] = float(du @ dr / (du.norm() * dr.norm() + 1e-30))
            num += float(du @ dr)
            d1 += float(du @ du)
            d2 += float(dr @ dr)
    return per, num / math.sqrt(d1 * d2 + 1e-300)


def check_iterations(res, g, strict_tol=2e-5, per_tensor_cos_min=0.98, f64=None, verbose=False):
    """iteration 0 element-wise; iteration 1 through the direction of its parameter update: the stem's float32
    gradients jump by percents when a 1e-6 weight difference flips a ReLU or a pooling arg-max, and Adam's
    normalisation (first steps ~ lr * sign(g)) turns that into O(lr) differences.  Measured on the reference itself:
    its float32 run against the float64 evaluation of the same two iterations has update cosines of 0.929
    (conv1.weight, 144 elements) ... 0.98 per tensor and 0.987 over all parameters.  So:
      * per_tensor_cos_min (the CPU oracle: same ATen kernels as the reference, hence the same flips): every tensor;
      * f64 (an implementation with its own float32 rounding, i.e. the GPU): a tensor-by-tensor comparison of two
        float32 runs only measures those flips.  Instead the run must be as close to the float64 trace as the
        reference's own float32 run is: cosine over ALL parameters >= the reference's - 0.02, and every tensor of
        >= 2048 elements >= the reference's - 0.05 (smaller tensors of >= 64 elements: - 0.15)."""
    lr = float(g["lr"])
    loss0, sd0, ema0 = res[0]
    assert abs(loss0 - float(g["it0.loss"])) < 1e-4 * abs(float(g["it0.loss"]))
    for k, v in sd0.items():
        ref = torch.from_numpy(g[f"it0.model.{k}"])
        if v.dtype == torch.int64:
            assert torch.equal(v, ref), k
            continue
        v, ref = _key_bias_free(k, v.flatten()), _key_bias_free(k, ref.flatten())
        tol = strict_tol + 1e-5 * ref.abs().max().item()
        flips = int(((v - ref).abs() > tol).sum())      # elements whose ~zero gradient changed sign between the two runs
        assert flips <= max(1, int(0.005 * v.numel())), (k, flips, (v - ref).abs().max().item())
        assert (v - ref).abs().max().item() <= 2.2 * lr + tol, k
    for k, v in ema0.items():
        ref = torch.from_numpy(g[f"it0.ema.{k}"])
        if v.dtype == torch.int64:
            assert torch.equal(v, ref), k
            continue
        assert (v - ref).abs().max().item() <= 2.2 * lr + strict_tol + 1e-5 * ref.abs().max().item(), k   # decay(0) = 0.1: ema = 0.1 ema + 0.9 model
    loss1, sd1, ema1 = res[1]
    assert abs(loss1 - float(g["it1.loss"])) < 1e-3 * abs(float(g["it1.loss"]))
    gold0 = {k: torch.from_numpy(g[f"it0.model.{k}"]) for k in sd1}
    gold1 = {k: torch.from_numpy(g[f"it1.model.{k}"]) for k in sd1}
    for k, v in sd1.items():
        ref1 = gold1[k]
        if v.dtype == torch.int64:
            assert torch.equal(v, ref1), k
            continue
        assert (v - ref1).abs().max().item() <= 4.4 * lr + 1e-4, k
        if "running_" in k or k == "pos_embed":
            assert torch.allclose(v, ref1, rtol=2e-2, atol=2e-3), k   # statistics of activations whose weights moved by O(lr)
    if per_tensor_cos_min is not None:
        per, _ = _update_cosines(sd1, sd0, gold1, gold0)
        for k, c in per.items():
            assert c > per_tensor_cos_min, (k, c)
    if f64 is not None:
        per_run, glob_run = _update_cosines(sd1, sd0, f64[1], f64[0])
        per_ref, glob_ref = _update_cosines(gold1, gold0, f64[1], f64[0])
        if verbose:
            print(f"update cosine vs float64 trace, all parameters: this run {glob_run:.4f}, reference float32 {glob_ref:.4f}")
            for k in per_run:
                if per_run[k] < 0.99 or per_ref[k] < 0.99:
                    print(f"  {k:48s} {sd1[k].numel():7d}  run {per_run[k]:.4f}  reference {per_ref[k]:.4f}")
        assert glob_run >= glob_ref - 0.02, (glob_run, glob_ref)
        for k, c in per_run.items():
            if sd1[k].numel() >= 2048:
                assert c >= per_ref[k] - 0.05, (k, c, per_ref[k])
            else:
                # small tensors (conv1.weight, BatchNorm / LayerNorm affine parameters, biases; >= 64 elements): a looser
                # floor, but a floor -- a sign or ordering bug confined to one of them in the second AdamW step would
                # turn its update around (cosine near 0 or negative), which rounding-level flips never do
                assert c >= per_ref[k] - 0.15, (k, c, per_ref[k])
    for k, v in ema1.items():
        ref = torch.from_numpy(g[f"it1.ema.{k}"])
        if v.dtype == torch.int64:
            assert torch.equal(v, ref), k
            continue
        assert (v - ref).abs().max().item() <= 4.4 * lr + 1e-4, k


def test_sam_adamw_ema_iterations_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_iter.npz"))
    check_iterations(run_oracle_iterations(g), g)


def test_greedy_decode_matches_reference_converter(golden_dir):
    g = np.load(os.path.join(golden_dir, "greedy_decode.npz"))
    seqs = O.greedy_decode(g["logits"])
    lens = g["lens"]
    assert [len(s) for s in seqs] == lens.tolist()
    assert np.array_equal(np.concatenate([np.asarray(s, np.int32) for s in seqs]), g["flat"])
