#!/usr/bin/env python3
"""Headline benchmark: line-images/s of one HTR-VT training step
(forward + CTC + backward [+ RCCL gradient all-reduce] + AdamW) on synthetic
64x1024 grey line images, batch 128 per GPU, bf16 MFMA path.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...            # no launcher: spawns the N rank processes itself (before any GPU call)
    python bench.py --gpus N --scaling strong   # global batch 128 split over the ranks (default: weak, 128 per GPU)

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     -- the dominant MFMA kernel: algorithmic FLOPs / HIP-event duration, measured live
  cpu_baseline -- the CPU oracle (a port of the reference's PyTorch CPU path) timed on this host
"""
import argparse
import json
import os
import sys
import time

# Before the HIP runtime starts: the step uses four streams of its own (main, weight gradients, weight packs, gradient
# collectives) and RCCL adds more; with HIP's default of 4 hardware queues per process the main and the weight-gradient
# stream of a data-parallel run landed on the SAME queue and ran one after the other (43.8 ms per step instead of 41.0).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "split_bf16": 2500.0}        # dense MFMA peaks, MI355X_MICROARCH.md (split: MFMA FLOPs = 3x algorithmic)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU (weak scaling) / in total (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch images per GPU; strong: --batch images in total, split evenly over the ranks")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "split_bf16"], help="bf16: the throughput path; f32 / split_bf16: the two parity paths (f32-input MFMA / hi+lo bf16 operands)")
    # model size: create_model's constants (HTR_VT.py:244-254) by default; BASELINE configs 1 / 5 name other sizes, which
    # the reference reaches through MaskedAutoencoderViT(...) directly (HTR_VT.py:143-151)
    ap.add_argument("--embed-dim", type=int, default=768)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--heads", type=int, default=6)
    ap.add_argument("--nb-cls", type=int, default=80)
    ap.add_argument("--forward-only", action="store_true", help="config 2: eval forward + CTC loss only")
    ap.add_argument("--sam", action="store_true", help="the reference's full iteration (train.py:119-128): SAM(AdamW) = two "
                    "fwd+bwd passes + climb/restore + AdamW + ModelEma update; images/s counts each image once")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend. nccl (= RCCL) is the measured one; gloo lets N ranks SHARE one GPU (rank r uses "
                         "device r %% device_count) to rehearse the N > 1 control flow of this script on a one-GPU box -- the "
                         "numbers of such a run mean nothing")
    ap.add_argument("--rehearse-collectives", action="store_true", help="N = 1 under a launcher: run the RCCL gradient "
                    "all-reduces anyway (exercises the N > 1 code path on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-path", action="store_true", help="skip the short float32 parity-path timing beside the bf16 line")
    ap.add_argument("--no-fuse-bn", action="store_true", help="A/B: separate BatchNorm-backward reduction pass")
    ap.add_argument("--no-overlap-wgrad", action="store_true", help="A/B: weight gradients on the main stream")
    ap.add_argument("--no-fuse-stem", action="store_true", help="A/B: conv1 tensor materialised (conv1_fwd + bn_relu_maxpool) "
                    "instead of the one-pass stem forward (the bf16 default; the float32 parity path never fuses unless --fuse-stem)")
    ap.add_argument("--fuse-stem", action="store_true", help="one-pass stem forward also on the float32 path")
    ap.add_argument("--deterministic", action="store_true", help="A/B: split-K weight gradients through ordered slabs instead "
                    "of float atomics also on the bf16 path (the float32 path always does)")
    ap.add_argument("--parallel-classes", action="store_true", help="A/B: parity-class dgrad launches of a strided conv on separate streams")
    ap.add_argument("--no-fused-attention", action="store_true", help="A/B: batched GEMMs + row softmax instead of csrc/attention.hip")
    ap.add_argument("--gemm-table", default=None, help="write the per-shape MFMA launch table (roofline leg) to this file")
    ap.add_argument("--traffic-json", default=None, help="PMC traffic summary (tools/collect_pmc.sh -> tools/pmc_summary.py) "
                    "the roofline's `traffic` is read from; default: the newest profiles/r*_pmc_traffic.json")
    ap.add_argument("--graph", default="off", choices=["off", "on", "multi"], help="replay the training step as ONE captured HIP "
                    "graph (Trainer.capture_step) inside the timed region instead of ~360 eager launches; on: captured on one "
                    "stream, multi: with the eager step's side streams (weight gradients, weight packs)")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1: skip the strong-scaling sub-record (global batch "
                    "128 split over the ranks; eager launches unless --graph on / multi) that follows the timed weak-scaling steps")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-iters", type=int, default=4, help="timed CPU iterations (about 10 s of host work in total)")
    return ap.parse_args()


def synthetic_batch(B, H, W, nb_cls, N, seed=0):
    """SURVEY.md 8(d) synthetic inputs: uniform [0,1) images, CTC-feasible random targets (lengths 20 .. min(90, N/2))"""
    import numpy as np
    g = torch.Generator().manual_seed(1234 + seed)
    x = torch.rand(B, 1, H, W, generator=g)
    rng = np.random.default_rng(seed)
    hi = max(3, min(90, N // 2))
    lo = min(20, hi - 1)
    lengths = rng.integers(lo, hi, size=B).astype(np.int32)
    targets = rng.integers(1, nb_cls, size=int(lengths.sum())).astype(np.int32)
    return x, targets, lengths


def cpu_baseline(args, mask):
    """oracle (CPU port of the reference path) on a bounded sample of the same workload"""
    from oracle import htrvt_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))        # a 1-GPU box share is ~16 cores; more threads than cores only thrash
    torch.set_num_threads(cores)
    cfg = O.Config(args.nb_cls, (64, args.width), embed_dim=args.embed_dim, depth=args.depth, num_heads=args.heads)
    sd = O.init_state_dict(cfg, seed=123)
    x, tg, tl = synthetic_batch(args.cpu_batch, 64, args.width, args.nb_cls, cfg.num_patches, seed=0)
    times = []
    for it in range(args.cpu_iters + 1):
        t0 = time.perf_counter()
        if args.forward_only:
            with torch.no_grad():
                y = O.forward(sd, cfg, x, train=False)
                y.permute(1, 0, 2).log_softmax(2)
        else:
            O.loss_and_grads(sd, cfg, x, tg, tl, keep_mask=mask, train=True)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / len(times[1:])
    out = {"value": round(args.cpu_batch / t, 3), "unit": "line-images/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
           "sample": f"{args.cpu_iters} timed iterations (1 warm-up) of batch {args.cpu_batch} 64x{args.width}, "
                     f"d{args.embed_dim}/{args.depth}L/{args.heads}h, "
                     f"{'eval forward + log_softmax' if args.forward_only else 'fwd+CTC+bwd (torch autograd over the oracle)'}"
                     f", torch CPU float32, {cores} threads"}
    # SURVEY.md 8(d)'s CPU case beside it: BASELINE config 1 (the reference's own CPU-runnable shape, run/iam.sh: 64x512,
    # batch 8) at d256/4L/4h and at create_model's d768/4L/6h, training step, a few seconds of host work
    cfg1 = {}
    for tag, (D_, L_, h_) in (("d256_4L_4h", (256, 4, 4)), ("d768_4L_6h", (768, 4, 6))):
        c1 = O.Config(80, (64, 512), embed_dim=D_, depth=L_, num_heads=h_)
        sd1 = O.init_state_dict(c1, seed=123)
        x1, tg1, tl1 = synthetic_batch(8, 64, 512, 80, c1.num_patches, seed=0)
        ts = []
        for it in range(3):
            t0 = time.perf_counter()
            O.loss_and_grads(sd1, c1, x1, tg1, tl1, keep_mask=None, train=True)
            ts.append(time.perf_counter() - t0)
        cfg1[tag] = round(8 / (sum(ts[1:]) / 2), 2)
    out["config1_64x512_b8_train_images_per_s"] = cfg1
    return out


def parity_path(args, dev, x, tg, tl, keep):
    """the same training step on the two paths that meet BASELINE.json's 1e-3 logit gate, timed briefly beside the bf16
    headline so that the driver's record carries them; reported, not part of `value`:
      split_bf16 -- float32 activations, conv / Linear operands split into hi + lo bf16 parts on the bf16 matrix cores
                    (three products, float32 accumulate; csrc/split.hip)
      f32        -- f32-input MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate), bitwise-reproducible reductions"""
    from htrvt_amd.trainer import Trainer
    out = {}
    for tag, cd in (("split_bf16", "split_bf16"), ("f32", torch.float32)):
        torch.manual_seed(123)
        model = build_model(args, cd).to(dev).train()
        tr = Trainer(model, max_lr=1e-3, weight_decay=0.5, world_size=1, use_collectives=False)
        steps = 3
        for _ in range(3):      # warm-up: the caching allocator's pool reaches its steady-state size by the third split step (both legs: same step's loss)
            tr.step(x, tg, tl, keep_mask=keep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.step(x, tg, tl, keep_mask=keep)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[tag] = {"ms_per_step": round(dt * 1e3, 2), "value": round(x.shape[0] / dt, 1), "unit": "line-images/s", "steps": steps,
                    "loss": float(loss.item())}
        del tr, model
        torch.cuda.empty_cache()
    best = out["split_bf16"]
    return {"dtype": "split-bf16 (f32 activations, hi+lo bf16 MFMA operands)", **best, "f32_mfma": out["f32"],
            "note": "paths whose logits are within 1e-3 of the reference (tests/test_full_shape_gpu.py), same step and batch"}


def set_stem_fusion(eng, args):
    """the engine's default (fused on bf16, two-kernel form on the float32 parity path) unless a flag overrides it"""
    if args.no_fuse_stem:
        eng.fuse_stem_forward = False
    elif args.fuse_stem:
        eng.fuse_stem_forward = True


def build_model(args, dtype):
    """create_model(nb_cls, [64, W]) (HTR_VT.py:244-254) -- or, for another width / depth / head count, the constructor it
    wraps with the same patch size, MLP ratio and LayerNorm eps"""
    from functools import partial
    from htrvt_amd.model import HTR_VT
    if (args.embed_dim, args.depth, args.heads) == (768, 4, 6):
        return HTR_VT.create_model(nb_cls=args.nb_cls, img_size=[64, args.width], compute_dtype=dtype)
    return HTR_VT.MaskedAutoencoderViT(args.nb_cls, img_size=[64, args.width], patch_size=(4, 64), embed_dim=args.embed_dim,
                                       depth=args.depth, num_heads=args.heads, mlp_ratio=4,
                                       norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), compute_dtype=dtype)


def newest_traffic_json():
    """profiles/rNN_pmc_traffic.json of the highest round (written by tools/collect_pmc.sh + tools/pmc_summary.py)"""
    import glob
    import re
    best = None
    for pth in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")):
        mt = re.match(r"r(\d+)_pmc_traffic\.json$", os.path.basename(pth))
        if mt and (best is None or int(mt.group(1)) > best[0]):
            best = (int(mt.group(1)), pth)
    return best[1] if best else None


def strong_leg(args, dev, world, rank, dtype, use_dist, steps=10, warmup=3):
    """global batch `--batch` split evenly over the ranks, eager launches unless --graph on / multi asks for the step replayed as
    one HIP graph (and then eager again if the capture is refused), barrier + synchronize on both sides, max over ranks --
    the same protocol as the main timed region"""
    from htrvt_amd.trainer import Trainer
    Bs = args.batch // world
    torch.manual_seed(123)
    model = build_model(args, dtype).to(dev).train()
    N = model.num_patches
    x, tg, tl = synthetic_batch(Bs, 64, args.width, args.nb_cls, N, seed=rank)
    x = x.to(dev)
    torch.manual_seed(7)
    keep = model.generate_span_mask(N, 0.4, 8)
    tr = Trainer(model, max_lr=1e-3, weight_decay=0.5, world_size=world, use_collectives=use_dist and world > 1)
    tr.step(x, tg, tl, keep_mask=keep)
    # eager launches by default: on this runtime a replayed graph of the step is SLOWER than its ~360 eager launches (one
    # MI355X, 16 images: 8.0 ms replayed from a one-stream capture, 14.3 ms from the four-stream capture, 6.9 ms eager --
    # hipGraphLaunch spaces the kernel nodes further apart than back-to-back launches do)
    graphed, note = args.graph != "off", None

    def one():
        return tr.step(x, tg, tl, keep_mask=keep)
    if graphed:
        try:
            gs = tr.capture_step(x, max_target_len=max(int(tl.max()), 1), masked=True, single_stream=args.graph == "on")

            def one():       # noqa: F811
                return gs.step(gs.img, tg, tl, keep_mask=keep)
        except Exception as e:      # noqa: BLE001 -- a refused capture (e.g. a collective that cannot be recorded) is reported, not fatal
            graphed, note = False, f"graph capture refused ({type(e).__name__}: {str(e)[:120]}); eager launches"
    # every rank takes the same branch: a refused capture on one rank would desynchronise the collectives
    if use_dist and args.graph != "off":
        flag = torch.tensor([1 if graphed else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if graphed and int(flag.item()) == 0:
            graphed, note = False, "graph capture refused on another rank; eager launches"

            def one():       # noqa: F811
                return tr.step(x, tg, tl, keep_mask=keep)
    for _ in range(warmup):
        one()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt_ = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt_], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_ = float(t.item())
    out = {"global_batch": Bs * world, "batch_per_gpu": Bs, "steps": steps, "warmup": warmup, "ms_per_step": round(dt_ / steps * 1e3, 3),
           "value": round(Bs * world * steps / dt_, 1), "unit": "line-images/s", "graph_replay": graphed}
    if note:
        out["note"] = note
    del tr, model
    torch.cuda.empty_cache()
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves.  This process has made no
    HIP call yet and never will (it only waits), each rank is a fresh interpreter with the torch.distributed.run
    environment variables; rank 0's JSON line goes to our stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        rc = p_.wait() or rc
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.backend == "gloo":
        local %= max(torch.cuda.device_count(), 1)      # rehearsal: the ranks share the devices that exist
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # one rank needs no collective: under a launcher at N = 1 the process group is still created (the launcher's
    # contract), but gradients are not "all-reduced" with themselves (RCCL's one-rank all-reduce of the 214 MB buffer
    # costs 2.4 ms per step) unless --rehearse-collectives asks for the full N > 1 code path on one GPU
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import htrvt_amd
    from htrvt_amd import ops
    from htrvt_amd.ctc import ctc_forward_backward
    from htrvt_amd.model import HTR_VT
    from htrvt_amd.trainer import Trainer

    dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "split_bf16": "split_bf16"}[args.dtype]
    torch.manual_seed(123)                                                  # option.py default --seed 123
    model = build_model(args, dtype).to(dev)
    N = model.num_patches
    B = args.batch
    if args.scaling == "strong":
        assert args.batch % world == 0, f"strong scaling: --batch {args.batch} must divide over {world} ranks"
        B = args.batch // world
    x, tg, tl = synthetic_batch(B, 64, args.width, args.nb_cls, N, seed=rank)
    x = x.to(dev)
    torch.manual_seed(7)
    keep = model.generate_span_mask(N, 0.4, 8)                              # run/iam.sh: --mask-ratio 0.4 --max-span-length 8

    if args.forward_only:
        model.eval()
        eng = model._engine(dev)
        set_stem_fusion(eng, args)
        P = dict(model.state_dict(keep_vars=True))

        def one_step():
            y = eng.forward(P, x, train=False, save=False)
            nll, _ = ctc_forward_backward(y, tg, tl, want_grad=False)
            return nll
    else:
        model.train()
        tr = Trainer(model, max_lr=1e-3, weight_decay=0.5, world_size=world,
                     use_collectives=use_dist and (world > 1 or args.rehearse_collectives))
        tr.engine.fuse_bn_backward = not args.no_fuse_bn
        tr.engine.overlap_wgrad = not args.no_overlap_wgrad
        set_stem_fusion(tr.engine, args)
        tr.engine.deterministic = tr.engine.deterministic or args.deterministic
        tr.engine.fused_attention = not args.no_fused_attention
        tr.engine.parallel_classes = args.parallel_classes

        if args.sam:
            from htrvt_amd.ema import ModelEma
            ema = ModelEma(model, 0.9999)
            torch.manual_seed(8)
            keep2 = model.generate_span_mask(N, 0.4, 8)
            it = [0]

            def one_step():
                l_ = tr.sam_step(x, tg, tl, keep_mask=keep, keep_mask2=keep2)
                ema.update(model, num_updates=it[0] / 2)
                it[0] += 1
                return l_
        elif args.graph != "off":
            tr.step(x, tg, tl, keep_mask=keep)          # lazily sized workspaces, one-time kernel attributes
            gs = None
            try:
                gs = tr.capture_step(x, max_target_len=max(int(tl.max()), 1), masked=True, single_stream=args.graph == "on")
            except Exception as e:      # noqa: BLE001 -- a refused capture (gloo, a collective that cannot be recorded): eager, and say so
                print(f"bench.py: graph capture refused ({type(e).__name__}: {str(e)[:160]}); eager launches", file=sys.stderr)
            if use_dist:                # every rank takes the same branch
                flag = torch.tensor([1 if gs is not None else 0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    gs = None
            if gs is None:
                args.graph = "off"

            def one_step():
                return gs.step(gs.img, tg, tl, keep_mask=keep) if gs is not None else tr.step(x, tg, tl, keep_mask=keep)
        else:
            def one_step():
                return tr.step(x, tg, tl, keep_mask=keep)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    # ---- strong-scaling leg (N > 1, after the timed weak-scaling steps; not part of `value`): the global batch of 128
    # split over the ranks -- SURVEY.md 8(d) asks for both curves, the driver's one command records both ----
    strong = None
    if world > 1 and args.scaling == "weak" and not (args.forward_only or args.sam or args.no_strong_leg) and args.batch % world == 0:
        strong = strong_leg(args, dev, world, rank, dtype, use_dist)
    elif world == 1 and args.scaling == "weak" and not (args.forward_only or args.sam):
        strong = {"global_batch": B, "batch_per_gpu": B, "ms_per_step": round(ms, 3), "value": round(value, 1),
                  "unit": "line-images/s", "note": "one GPU: the strong- and the weak-scaling step are the same step"}

    # ---- roofline leg: HIP events around every MFMA launch of two extra steps (not part of `value`) ----
    roof = None
    # every rank runs the two extra steps (they contain collectives); only rank 0 records events
    if rank == 0:
        ops.PROFILE = {}
    if not args.forward_only:
        saved_overlap, tr.engine.overlap_wgrad = tr.engine.overlap_wgrad, False   # one stream: events time ONE kernel
    # (a replayed graph has no per-launch events: the two profiled steps always run as eager launches)
    prof_step = one_step if (args.forward_only or args.sam or args.graph == "off") else (lambda: tr.step(x, tg, tl, keep_mask=keep))
    for _ in range(2):
        prof_step()
    torch.cuda.synchronize()
    if not args.forward_only:
        tr.engine.overlap_wgrad = saved_overlap
    if rank == 0:
        prof, ops.PROFILE = ops.PROFILE, None
        rows, by_sym = [], {}
        for key, ent in prof.items():
            ts = [a.elapsed_time(b) for a, b in ent["events"]]
            rows.append((sum(ts), key, ent["flops"], sum(ts) / len(ts), len(ts) // 2, ent["kernel"]))
            sym = by_sym.setdefault(ent["kernel"], {"ms": 0.0, "flops": 0.0, "n": 0, "shapes": []})
            sym["ms"] += sum(ts)
            sym["flops"] += ent["flops"] * len(ts)
            sym["n"] += len(ts)
            sym["shapes"].append((sum(ts), key, ent["flops"], sum(ts) / len(ts), len(ts) // 2))
        rows.sort(reverse=True)
        names = {0: "plain", 1: "conv-fwd", 2: "conv-dgrad", 3: "conv-wgrad"}
        if args.gemm_table:
            with open(args.gemm_table, "w") as f:
                f.write("ms/step  avg_us  n/step  TFLOP/s  kernel | dtype,aL,bL,gather,M,N,K,batch\n")
                for t_, k_, fl_, av_, n_, sy_ in rows:
                    f.write(f"{t_ / 2:7.3f} {av_ * 1e3:8.1f} {n_:4d} {fl_ / (av_ * 1e-3) / 1e12:8.1f}  {sy_} | {k_}\n")
        # dominant kernel = the rocprofv3 SYMBOL with the largest time share (one symbol serves several launch shapes);
        # achieved = its algorithmic FLOPs per launch / its average launch duration, both averaged over all its launches
        # of a step, which is what the `rocprofv3 --kernel-trace --stats` row of that symbol averages too
        symname, sym = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = sym["ms"] / sym["n"]
        flops = sym["flops"] / sym["n"]
        per_step = sym["n"] // 2
        achieved = flops / (avg_ms * 1e-3) / 1e12
        sym["shapes"].sort(reverse=True)
        shapes = [{"shape": f"{names[k_[3]]} M={k_[4]} N={k_[5]} K={k_[6]} batch={k_[7]}", "launches_per_step": n_,
                   "avg_ms": round(av_, 4), "tflops": round(fl_ / (av_ * 1e-3) / 1e12, 1)} for _, k_, fl_, av_, n_ in sym["shapes"][:4]]
        traffic, traffic_src = None, None
        try:   # HBM bytes per launch (mean over the symbol's launches) from the newest committed PMC passes
            tj = args.traffic_json or newest_traffic_json()
            with open(tj) as f:
                pm = json.load(f)
            ent = pm["symbols"].get(symname)
            if ent is not None and B == 128 and args.width == 1024 and args.dtype == "bf16" and not args.forward_only:
                traffic, traffic_src = ent["traffic_bytes_per_launch_mean"], os.path.relpath(tj, ROOT)
        except (OSError, KeyError, ValueError, TypeError):
            pass
        roof = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS[args.dtype], 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": symname, "launches_per_step": per_step, "avg_ms": round(avg_ms, 4),
                "algorithmic_gflop_per_launch": round(flops / 1e9, 1), "ms_per_step": round(sym["ms"] / 2, 3),
                "top_shapes": shapes,
                "mfma_ms_per_step": round(sum(r[0] for r in rows) / 2, 2),
                "all_mfma_tflops": round(sum(r[2] * len(prof[r[1]]["events"]) for r in rows) / sum(r[0] for r in rows) / 1e9, 1)}

    if rank == 0:
        model_tag = f"d{args.embed_dim}/{args.depth}L/{args.heads}h, nb_cls {args.nb_cls}"
        what = ("fwd+CTC (eval)" if args.forward_only else "SAM(AdamW) 2x(fwd+bwd+CTC)+EMA" if args.sam else "fwd+bwd+CTC")
        eng_ = eng if args.forward_only else tr.engine
        out = {"metric": f"line-images/sec (64x{args.width}, global B={B * world}, {B} per GPU) {what}",
               "value": round(value, 1), "unit": "line-images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": (f"HTR-VT ({model_tag}) eval forward + CTC loss" if args.forward_only else
                                        f"HTR-VT ({model_tag}) reference iteration: SAM(AdamW) 2x(fwd + CTC + bwd) + EMA"
                                        if args.sam else
                                        f"HTR-VT ({model_tag}) training step: fwd + fused CTC + bwd + AdamW"
                                        + (" + RCCL grad all-reduce" if world > 1 else "")),
                          "image": f"1x64x{args.width}", "batch_per_gpu": B, "global_batch": B * world,
                          "tokens_per_image": N, "mask": "span 0.4/8 (run/iam.sh)", "parallelism": f"dp{world}",
                          "engine_flags": {k: getattr(eng_, k) for k in ("fuse_stem_forward", "fuse_bn_backward", "overlap_wgrad",
                                                                          "fused_attention", "deterministic", "parallel_classes")},
                          "graph_replay": args.graph != "off" and not (args.forward_only or args.sam),
                          "loss": float(loss.mean().item()) if loss is not None else None},
               "roofline": roof}
        if world > 1:
            out["config"]["collective_backend"] = "rccl" if args.backend == "nccl" else "gloo (ranks share a GPU: a rehearsal of the control flow, not a measurement)"
        if strong is not None:
            out["strong"] = strong
        if (not args.no_parity_path and world == 1 and args.dtype == "bf16" and not args.forward_only and not args.sam
                and args.scaling == "weak"):
            out["parity_path"] = parity_path(args, dev, x, tg, tl, keep)
        if not args.no_cpu_baseline:     # rank 0, after the timed region (N > 1: the other ranks wait at the closing barrier)
            out["cpu_baseline"] = cpu_baseline(args, keep)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
