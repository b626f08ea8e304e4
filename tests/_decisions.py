"""Test helper: where two forward passes of the engine (float32 path / split-bf16 path) took different DISCRETE decisions --
the stem max-pool's arg-max (resnet18.py:77), every ReLU of the residual stages (resnet18.py:29,37) and the final max-pool's
arg-max (resnet18.py:82).  Input: the activations a forward pass saves for its backward (`y.grad_fn.saved_acts`)."""
import torch
import torch.nn.functional as F


def decision_flips(sv_a, sv_b):
    """{decision set: (elements that differ, elements)}"""
    out = {"stem max-pool arg-max": (int((sv_a["idx"] != sv_b["idx"]).sum()), sv_a["idx"].numel())}
    for ba, bb in zip(sv_a["stem_blocks"], sv_b["stem_blocks"]):
        for key in ("a1", "out"):
            out[f"{ba['p']} {key} ReLU"] = (int(((ba[key] > 0) != (bb[key] > 0)).sum()), ba[key].numel())

    def argmax(t):     # [B, Hc, Wc, D] -> indices of max_pool2d(3, stride (2, 1), padding 1)
        return F.max_pool2d(t.permute(0, 3, 1, 2).float(), 3, stride=(2, 1), padding=1, return_indices=True)[1]
    ia, ib = argmax(sv_a["l3"]), argmax(sv_b["l3"])
    out["final max-pool arg-max"] = (int((ia != ib).sum()), ia.numel())
    return out


def total(flips):
    return sum(v[0] for v in flips.values()), sum(v[1] for v in flips.values())


def rel_errors(got, ref):
    """(max-abs error / max-abs of the reference, relative L2 error) of two same-shaped arrays / tensors"""
    g, r = torch.as_tensor(got).double().flatten(), torch.as_tensor(ref).double().flatten()
    return float((g - r).abs().max() / max(float(r.abs().max()), 1e-6)), float((g - r).norm() / max(float(r.norm()), 1e-12))
