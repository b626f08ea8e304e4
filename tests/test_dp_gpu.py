"""BASELINE config 3 (data-parallel training step, RCCL gradient all-reduce) exercised on ONE MI355X: a fresh
process creates a 1-rank `nccl` (= RCCL) process group before its first kernel and runs Trainer(use_collectives=True)
-- the three bucketed all-reduces on their own stream, the waits on the engine's weight-gradient stream, the parameter /
buffer broadcast -- beside Trainer(use_collectives=False).  A SUM all-reduce over one rank is the identity, so
parameters, gradients and BatchNorm buffers after two AdamW steps and after one SAM(AdamW) iteration must agree BIT FOR
BIT on the float32 path (deterministic reductions); any missing stream dependency shows up as a difference.
SURVEY.md 8(e); the reference itself has no multi-GPU code (model_v1/train.py is single-process)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)      # before the first kernel of this process
from functools import partial
import htrvt_amd
from htrvt_amd.model import HTR_VT
from htrvt_amd.trainer import Trainer
from oracle import htrvt_oracle as O                 # synthetic inputs only (test infrastructure)

dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[2]]
D, L, h = 256, 4, 4
cfg = O.Config(80, (64, 512), embed_dim=D, depth=L, num_heads=h)
x, tg, tl = O.synthetic_batch(8, 64, 512, 80, cfg.num_patches, seed=3)
x = x.to(dev)


def build(coll):
    torch.manual_seed(123)
    m = HTR_VT.MaskedAutoencoderViT(80, img_size=[64, 512], patch_size=(4, 64), embed_dim=D, depth=L, num_heads=h,
                                    mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), compute_dtype=dtype)
    m = m.to(dev).train()
    tr = Trainer(m, max_lr=1e-3, weight_decay=0.5, world_size=1, use_collectives=coll)
    tr.engine.deterministic = True        # slab split-K also on bf16: bitwise comparable runs
    return m, tr


def run(coll):
    m, tr = build(coll)
    torch.manual_seed(7)
    k1 = m.generate_span_mask(cfg.num_patches, 0.4, 8)
    k2 = m.generate_span_mask(cfg.num_patches, 0.4, 8)
    out = []
    for _ in range(2):
        loss = tr.step(x, tg, tl, keep_mask=k1)
        out.append(tr.flat.flat_g.clone())
    out.append(tr.flat.flat_p.clone())
    tr.sam_step(x, tg, tl, keep_mask=k1, keep_mask2=k2)
    out.append(tr.flat.flat_g.clone())
    out.append(tr.flat.flat_p.clone())
    out += [b.clone() for _, b in m.named_buffers()]
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all()
    return out, (tr.flat.side is not None)


a, side_a = run(True)
b, side_b = run(False)
assert side_a and not side_b, "the collective run must own a collective stream"
bad = [i for i, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]
assert not bad, f"collective / plain runs differ in tensors {bad}"
assert float(a[0].abs().sum()) > 0
dist.barrier()
dist.destroy_process_group()
print("DP1 OK", len(a), "tensors bit-identical")
'''


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_trainer_collective_path_one_rank_nccl(dtype):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, dtype], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP1 OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


CHILD2 = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")                      # two ranks share ONE GPU: RCCL refuses that, gloo moves the buckets through the host
rank, world = dist.get_rank(), dist.get_world_size()
from functools import partial
import htrvt_amd
from htrvt_amd.model import HTR_VT
from htrvt_amd.trainer import Trainer
from oracle import htrvt_oracle as O                 # synthetic inputs only (test infrastructure)

D, L, h = 256, 4, 4
cfg = O.Config(80, (64, 512), embed_dim=D, depth=L, num_heads=h)
x, tg, tl = O.synthetic_batch(4, 64, 512, 80, cfg.num_patches, seed=11 + rank)      # every rank its own lines
x = x.to(dev)


def build(coll):
    torch.manual_seed(123)
    m = HTR_VT.MaskedAutoencoderViT(80, img_size=[64, 512], patch_size=(4, 64), embed_dim=D, depth=L, num_heads=h,
                                    mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), compute_dtype=torch.float32)
    m = m.to(dev).train()
    return m, Trainer(m, max_lr=1e-3, weight_decay=0.5, world_size=world if coll else 1, use_collectives=coll)


torch.manual_seed(7)
m, tr = build(True)
k1 = m.generate_span_mask(cfg.num_patches, 0.4, 8)
k2 = m.generate_span_mask(cfg.num_patches, 0.4, 8)
tr.step(x, tg, tl, keep_mask=k1)
g_dp = tr.flat.flat_g.clone()
tr.step(x, tg, tl, keep_mask=k1)
tr.sam_step(x, tg, tl, keep_mask=k1, keep_mask2=k2)
torch.cuda.synchronize()
p_dp = tr.flat.flat_p.clone()

# (1) the replicas stay in step: parameters after two AdamW steps and one SAM iteration are the same bits on both ranks
both = [torch.empty_like(p_dp) for _ in range(world)]
dist.all_gather(both, p_dp)
assert torch.equal(both[0], both[1]), int((both[0] != both[1]).sum())
assert torch.isfinite(p_dp).all()

# (2) what was applied is the MEAN gradient: the same first step without collectives on this rank's lines, averaged by hand
#     (1 / world is a power of two: scaling commutes with every rounding, so the comparison is bitwise)
m2, tr2 = build(False)
tr2.step(x, tg, tl, keep_mask=k1)
g_loc = tr2.flat.flat_g.clone()
dist.all_reduce(g_loc)
g_loc *= 1.0 / world
assert torch.equal(g_dp, g_loc), (int((g_dp != g_loc).sum()), float((g_dp - g_loc).abs().max()))
assert float(g_dp.abs().sum()) > 0
dist.barrier()
dist.destroy_process_group()
print("DP2 OK rank", rank)
'''


def test_trainer_two_ranks_on_one_gpu_gloo():
    """WORLD_SIZE = 2 through the real kernels: two processes on the one GPU of the test box, gloo as the transport (RCCL
    needs one device per rank).  Every rank trains on its own lines; the three bucketed all-reduces, the 1 / world gradient
    scale and the waits between the compute, weight-gradient and collective streams must leave (1) bit-identical parameters
    on both ranks after two AdamW steps and one SAM iteration and (2) exactly the mean of the two ranks' local gradients."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD2, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=600))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for rank, (pr, (so, se)) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and f"DP2 OK rank {rank}" in so, (rank, so[-2000:], se[-4000:])


def test_bench_two_rank_control_flow_rehearsal():
    """`bench.py --gpus 2` end to end on the one GPU of the test box (`--backend gloo`: the ranks share the device): the
    N > 1 branch of the script -- process group, per-rank data, barriers, max-over-ranks timing, the strong-scaling leg, the
    two profiled steps on every rank, rank 0's ONE JSON line -- which the driver's 8-GPU run is the first to execute
    otherwise.  The numbers of such a run mean nothing; the record's shape is what is checked."""
    import json
    env = dict(os.environ, GLOO_SOCKET_IFNAME="lo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--width", "512", "--embed-dim", "256", "--depth", "4", "--heads", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["roofline"]["achieved"] > 0
    assert d["strong"]["global_batch"] == 8 and d["strong"]["batch_per_gpu"] == 4 and d["strong"]["value"] > 0
    assert "gloo" in d["config"]["collective_backend"]
