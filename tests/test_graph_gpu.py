"""The training step as ONE replayed HIP graph (Trainer.capture_step / GraphedStep) against the eager step.

SURVEY.md 2b calls the ~360 launches of a step launch-bound at small per-rank batches (strong scaling, 8(d) / 8(e)); the
graph removes the host from the step.  It must not change a bit: the same kernels in the same order, per-step inputs
(labels, span mask, lr / step number of AdamW) refreshed through fixed device buffers.  Each test runs the SAME sequence of
steps -- different labels, masks and learning rates per step -- once eagerly and once as eager warm-up + replays, and
requires parameters, Adam moments, gradients, BatchNorm buffers and losses to be bit-identical."""
import os
import socket
import subprocess
import sys
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import htrvt_oracle as O          # synthetic inputs only

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trainer(cfg, dtype):
    from htrvt_amd.model import HTR_VT
    from htrvt_amd.trainer import Trainer
    sd = O.init_state_dict(cfg, seed=7, randomize_affine=True)
    m = HTR_VT.MaskedAutoencoderViT(cfg.nb_cls, img_size=[cfg.H, cfg.W], patch_size=cfg.patch, embed_dim=cfg.D,
                                    depth=cfg.depth, num_heads=cfg.heads, mlp_ratio=4,
                                    norm_layer=partial(nn.LayerNorm, eps=1e-6), compute_dtype=dtype)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    return m, Trainer(m, max_lr=1e-3, betas=(0.9, 0.99), weight_decay=0.5)


def _inputs(cfg, B, nsteps):
    out = []
    for it in range(nsteps):
        x, tg, tl = O.synthetic_batch(B, cfg.H, cfg.W, cfg.nb_cls, cfg.num_patches, seed=10 + it)
        torch.manual_seed(100 + it)
        keep = O.span_mask(cfg.num_patches, 0.4, 8)
        out.append((x.cuda(), tg, tl, keep, 1e-3 * (it + 1) / nsteps))
    return out


def _state(m, tr, losses):
    torch.cuda.synchronize()
    return dict(p=tr.flat.flat_p.clone(), g=tr.flat.flat_g.clone(), m=tr.flat_m.clone(), v=tr.flat_v.clone(),
                bufs=[b.clone() for _, b in m.named_buffers()], losses=[float(l_) for l_ in losses])


@pytest.mark.parametrize("dtype,shape", [(torch.float32, (64, 2, 2, 512, 4)), (torch.float32, (256, 4, 4, 512, 8)),
                                         (torch.bfloat16, (256, 4, 4, 512, 8))])
def test_captured_step_is_bit_identical_to_eager(dtype, shape):
    D, depth, heads, W, B = shape
    cfg = O.Config(80, (64, W), embed_dim=D, depth=depth, num_heads=heads)
    steps = _inputs(cfg, B, 4)

    m, tr = _trainer(cfg, dtype)
    losses = [tr.step(x, tg, tl, keep_mask=k, lr=lr).clone() for x, tg, tl, k, lr in steps]
    eager = _state(m, tr, losses)
    del m, tr

    m, tr = _trainer(cfg, dtype)
    x, tg, tl, k, lr = steps[0]
    losses = [tr.step(x, tg, tl, keep_mask=k, lr=lr).clone()]
    gs = tr.capture_step(x, max_target_len=cfg.num_patches // 2, masked=True)
    for x, tg, tl, k, lr in steps[1:]:
        losses.append(gs.step(x, tg, tl, keep_mask=k, lr=lr).clone())
    graph = _state(m, tr, losses)

    assert eager["losses"] == graph["losses"], (eager["losses"], graph["losses"])
    for key in ("g", "p", "m", "v"):
        assert torch.equal(eager[key], graph[key]), (key, int((eager[key] != graph[key]).sum()))
    for a, b in zip(eager["bufs"], graph["bufs"]):
        assert torch.equal(a, b)
    # and the model the graph trained is usable outside it: the eager forward re-packs from the updated weights
    m.eval()
    with torch.no_grad():
        y = m(steps[0][0])
    assert torch.isfinite(y).all()


def test_captured_step_rejects_what_it_was_not_captured_for():
    cfg = O.Config(80, (64, 512), embed_dim=64, depth=2, num_heads=2)
    steps = _inputs(cfg, 4, 1)
    m, tr = _trainer(cfg, torch.float32)
    x, tg, tl, k, lr = steps[0]
    with pytest.raises(RuntimeError):
        tr.capture_step(x, max_target_len=8)          # no eager step yet
    tr.step(x, tg, tl, keep_mask=k, lr=lr)
    gs = tr.capture_step(x, max_target_len=8, masked=True)
    with pytest.raises(ValueError):
        gs.step(x, tg, tl, keep_mask=k)               # labels longer than the captured bound
    short = np.minimum(tl, 8).astype(np.int32)
    with pytest.raises(ValueError):
        gs.step(x, tg[:int(short.sum())], short, keep_mask=None)    # captured WITH a span mask


CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
from functools import partial
import htrvt_amd
from htrvt_amd.model import HTR_VT
from htrvt_amd.trainer import Trainer
from oracle import htrvt_oracle as O

cfg = O.Config(80, (64, 512), embed_dim=256, depth=4, num_heads=4)
x, tg, tl = O.synthetic_batch(8, 64, 512, 80, cfg.num_patches, seed=3)
x = x.to(dev)


def build():
    torch.manual_seed(123)
    m = HTR_VT.MaskedAutoencoderViT(80, img_size=[64, 512], patch_size=(4, 64), embed_dim=256, depth=4, num_heads=4,
                                    mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), compute_dtype=torch.bfloat16)
    m = m.to(dev).train()
    return m, Trainer(m, max_lr=1e-3, weight_decay=0.5, world_size=1, use_collectives=True)


torch.manual_seed(7)
keep = O.span_mask(cfg.num_patches, 0.4, 8)
m, tr = build()
for _ in range(3):
    tr.step(x, tg, tl, keep_mask=keep)
torch.cuda.synchronize()
a = [tr.flat.flat_p.clone(), tr.flat.flat_g.clone()]
m, tr = build()
tr.step(x, tg, tl, keep_mask=keep)
gs = tr.capture_step(x, max_target_len=cfg.num_patches // 2)     # the three bucketed RCCL all-reduces are captured too
for _ in range(2):
    gs.step(x, tg, tl, keep_mask=keep)
torch.cuda.synchronize()
b = [tr.flat.flat_p.clone(), tr.flat.flat_g.clone()]
assert all(torch.equal(u, v) for u, v in zip(a, b)), "captured collective step differs from the eager one"
dist.barrier()
dist.destroy_process_group()
print("GRAPH DP1 OK")
'''


def test_captured_step_with_rccl_collectives_one_rank():
    """the data-parallel step (process group present, three bucketed all-reduces on the collective stream) captured as one
    graph under a ONE-RANK nccl group, fresh process, against the eager collective step: bit-identical"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GRAPH DP1 OK" in r.stdout, (r.stdout[-2000:], r.stderr[:3000], r.stderr[-3000:])
