#!/usr/bin/env python3
"""Generate tests/golden/variants.npz by running the REFERENCE's variant modules on CPU:
  * /root/reference/model_window/model/HTR_VT.py  Block._attend + Attention.forward (relative-position bias, 1-D windows,
    cyclic shift, zero padding to a multiple of the window + key_padding_mask)
  * /root/reference/model_sgm_2/model/sgm_head.py SGMHead._cross_attend (kv LayerNorm + single-head cross-attention)
Runs only in the dev container (needs /root/reference); the fixture is data (inputs, parameters, outputs, gradients).
timm (absent here) is needed by the import only: the same in-memory Mlp / DropPath stand-in as tools/make_goldens.py.

    python tools/make_goldens_variants.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "tests", "golden")


def _load(name, path, extra_path):
    sys.path.insert(0, extra_path)          # `from model import resnet18` inside the reference file
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        sys.path.remove(extra_path)
        for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
            del sys.modules[k]


def main():
    from make_goldens import _install_timm_stub
    _install_timm_stub()
    torch.set_num_threads(4)
    win = _load("ref_window_htr_vt", "/root/reference/model_window/model/HTR_VT.py", "/root/reference/model_window")
    sgm = _load("ref_sgm_head", "/root/reference/model_sgm_2/model/sgm_head.py", "/root/reference/model_sgm_2")
    out = {}

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import variant_cases as VC
    t64 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))      # noqa: E731

    # ---- windowed / relative-bias self-attention: Block._attend on the block's normalised input -------------------
    for case in VC.WINDOW_CASES:
        tag, B, N, dim, heads, P, ws, shift = case
        inp = VC.window_inputs(case)
        blk = win.Block(dim, heads, P, mlp_ratio=4.0, qkv_bias=True, window_size=ws, shift_size=shift).double()
        a = blk.attn
        with torch.no_grad():
            a.qkv.weight.copy_(t64(inp["qkv_w"]))
            a.qkv.bias.copy_(t64(inp["qkv_b"]))
            a.proj.weight.copy_(t64(inp["proj_w"]))
            a.proj.bias.copy_(t64(inp["proj_b"]))
            a.relative_position_bias_table.copy_(t64(inp["table"]))
        x = t64(inp["x"]).requires_grad_(True)
        y = blk._attend(x)
        y.backward(t64(inp["gout"]))
        out.update({f"win.{tag}.y": y.detach().numpy(), f"win.{tag}.dx": x.grad.numpy(),
                    f"win.{tag}.dtable": a.relative_position_bias_table.grad.numpy(),
                    f"win.{tag}.dqkv_b": a.qkv.bias.grad.numpy()})
        print("window", tag, tuple(y.shape), float(y.detach().abs().max()))

    # ---- SGM head cross-attention (kv LayerNorm + softmax(Q K^T / sqrt(D)) K) --------------------------------------
    for case in VC.SGM_CASES:
        tag, Bq, L, N, D = case
        inp = VC.sgm_inputs(case)
        head = sgm.SGMHead(d_vis=D, vocab_size_sgm=20, d_txt=32).double().eval()       # eval: Dropout is the identity
        with torch.no_grad():
            head.kv_norm.weight.copy_(t64(inp["ln_w"]))
            head.kv_norm.bias.copy_(t64(inp["ln_b"]))
        Q, Fv = t64(inp["Q"]).requires_grad_(True), t64(inp["F"]).requires_grad_(True)
        y = head._cross_attend(Q, Fv)
        y.backward(t64(inp["gout"]))
        out.update({f"sgm.{tag}.y": y.detach().numpy(), f"sgm.{tag}.dQ": Q.grad.numpy(), f"sgm.{tag}.dF": Fv.grad.numpy(),
                    f"sgm.{tag}.dln_w": head.kv_norm.weight.grad.numpy()})
        print("sgm", tag, tuple(y.shape), float(y.detach().abs().max()))

    out = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in out.items()}
    np.savez_compressed(os.path.join(OUT, "variants.npz"), **out)
    print("wrote", os.path.join(OUT, "variants.npz"), os.path.getsize(os.path.join(OUT, "variants.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
