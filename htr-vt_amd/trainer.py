"""Data-parallel training step of the HTR-VT hot path (one process per MI355X).

step = forward + fused CTC + backward (+ gradient all-reduce over RCCL/xGMI)
+ AdamW, i.e. the inner statements of /root/reference/model_v1/train.py:119-126
without the SAM second pass (SURVEY.md 8(f-1) lists SAM as a later row).

MI355X-first layout: every trainable parameter is a view into ONE flat float32
buffer (same for gradients and the two Adam moments), so the optimizer is one
kernel launch and the gradient exchange is three large all-reduces (xGMI is
point-to-point: few, large messages).  The encoder/head bucket -- the tail of
the flat buffer, complete early in backward -- and then the layer-3 bucket (78 %
of the stem's weights, complete two blocks later) are reduced on a side stream
while the rest of the stem backward still runs; only the last 22 MB (conv1,
layer1, layer2) are exchanged after the backward.
The data-parallel average is folded into the loss gradient (CTC grad_scale =
1/world_size), so the all-reduce is a plain SUM and nothing rescales afterwards.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import mark_weights_dirty, trace


class FlatParams:
    """Flat float32 parameter / gradient storage + the two gradient buckets.  Device-agnostic (the
    world-size-2 gloo test drives it on CPU); on CUDA the encoder bucket is reduced on a side stream."""

    def __init__(self, model, world_size=1, use_collectives=None):
        self.model = model
        self.world = world_size
        # collectives run whenever a process group exists (also at world_size 1, which rehearses the RCCL path)
        self.coll = (world_size > 1) if use_collectives is None else bool(use_collectives)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        dev = named[0][1].device
        total = sum((p.numel() + 3) // 4 * 4 for _, p in named)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.G = {}
        off, self.enc_start, self.l3_start = 0, None, None
        for n, p in named:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p)
            self.G[n] = self.flat_g[off:off + k].view_as(p)
            if self.enc_start is None and n.startswith("blocks."):
                self.enc_start = off        # blocks.*, norm.*, head.* follow in state_dict order
            if self.l3_start is None and n.startswith("patch_embed.layer3."):
                self.l3_start = off         # layer3.* is the tail of the stem segment: 19.5 M of its 25 M weights
            off += (k + 3) // 4 * 4
        if self.enc_start is None:
            self.enc_start = 0
        if self.l3_start is None or self.l3_start > self.enc_start:
            self.l3_start = self.enc_start
        self.side = torch.cuda.Stream(device=dev) if (self.coll and dev.type == "cuda") else None
        if self.coll:   # identical replicas: rank 0's parameters and BN buffers
            dist.broadcast(self.flat_p, 0)
            for _, b in model.named_buffers():
                dist.broadcast(b, 0)

    def check_views(self):
        base = self.flat_p.untyped_storage().data_ptr()
        for n, p in self.model.named_parameters():
            if p.requires_grad and p.data.untyped_storage().data_ptr() != base:
                raise RuntimeError(f"parameter {n} was re-bound outside the flat buffer; rebuild the Trainer")

    def _behind(self, producer):
        """the collective stream waits for everything enqueued so far on the current stream and on `producer` (the
        engine's weight-gradient stream): the bucket is complete then, and the main stream itself never stalls"""
        self.side.wait_stream(torch.cuda.current_stream())
        if producer is not None:
            self.side.wait_stream(producer)

    def reduce_encoder_bucket(self, producer=None):
        """all-reduce(SUM) of the blocks.*/norm/head gradients; on CUDA on the side stream, overlapping
        whatever the current stream enqueues next (the stem backward)."""
        if not self.coll:
            return
        if self.side is None:
            dist.all_reduce(self.flat_g[self.enc_start:])
            return
        self._behind(producer)
        with torch.cuda.stream(self.side):
            dist.all_reduce(self.flat_g[self.enc_start:])

    def reduce_layer3_bucket(self, producer=None):
        """all-reduce(SUM) of the patch_embed.layer3.* gradients, complete after the first two stem blocks of the
        backward; on CUDA on the side stream behind the encoder bucket, under the layer-2/1 backward."""
        if not self.coll or self.l3_start == self.enc_start:
            return
        if self.side is None:
            dist.all_reduce(self.flat_g[self.l3_start:self.enc_start])
            return
        self._behind(producer)
        with torch.cuda.stream(self.side):
            dist.all_reduce(self.flat_g[self.l3_start:self.enc_start])

    def reduce_stem_bucket(self, producer=None):
        """the rest of the stem (conv1, layer1, layer2: 5.5 M weights), after the backward: on the collective stream like
        the other two buckets (one stream = one RCCL queue, the three all-reduces never interleave), and the current
        stream then waits for all three -- the optimizer is the first consumer of any reduced gradient"""
        if not self.coll:
            return
        if self.side is None:
            dist.all_reduce(self.flat_g[:self.l3_start])
            return
        self._behind(producer)
        with torch.cuda.stream(self.side):
            dist.all_reduce(self.flat_g[:self.l3_start])
        torch.cuda.current_stream().wait_stream(self.side)


class Trainer:
    def __init__(self, model, max_lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.5, world_size=1,
                 use_collectives=None):
        from ._lib import lib  # noqa: F401  (fails loudly without the HIP library)
        self.model = model
        self.world = world_size
        self.lr, self.betas, self.eps, self.wd = max_lr, betas, eps, weight_decay
        self.step_count = 0
        dev = next(model.parameters()).device
        assert dev.type == "cuda", "Trainer needs the model on an MI355X (model.cuda())"
        self.flat = FlatParams(model, world_size, use_collectives)
        self.flat_m = torch.zeros_like(self.flat.flat_p)
        self.flat_v = torch.zeros_like(self.flat.flat_p)
        self.P = dict(model.state_dict(keep_vars=True))
        self.engine = model._engine(dev)

    def forward_backward(self, img, targets, lengths, keep_mask=None):
        """forward + CTC + backward on this rank's shard; the (already averaged) gradients land in the flat
        gradient buffer.  Returns the local mean loss (device scalar)."""
        from .ctc import stage_targets
        self.flat.check_views()     # a caller that re-bound p.data (reference SAM.second_step, model.to()) must rebuild the Trainer
        staged = stage_targets(targets, lengths, img.device)   # before the forward is enqueued (see stage_targets)
        return self._forward_backward_staged(img, staged, keep_mask)

    def _forward_backward_staged(self, img, staged, keep_mask):
        """the device work of forward_backward: label arrays already on the device (`staged`), keep_mask None or a host /
        device [N] tensor.  This is the body a captured step records (GraphedStep)."""
        from .ctc import ctc_forward_backward
        eng, fl = self.engine, self.flat
        fl.flat_g.zero_()
        with trace.range_("htrvt.forward"):
            y = eng.forward(self.P, img, keep_mask=keep_mask, train=True, save=True)
        with trace.range_("htrvt.ctc"):
            nll, dy = ctc_forward_backward(y, None, None, want_grad=True, grad_scale=1.0 / self.world, staged=staged)
        with trace.range_("htrvt.backward"):
            eng.backward(self.P, fl.G, dy, after_encoder=fl.reduce_encoder_bucket, after_layer3=fl.reduce_layer3_bucket)
            fl.reduce_stem_bucket()
        return nll.mean()

    def optimizer_step(self, lr=None):
        from ._lib import check, lib
        from .ops import ptr, stream
        self.step_count += 1
        fl = self.flat
        with trace.range_("htrvt.adamw"):
            check(lib.htrvt_adamw(ptr(fl.flat_p), ptr(fl.flat_g), ptr(self.flat_m), ptr(self.flat_v), fl.flat_p.numel(),
                                  float(self.lr if lr is None else lr), self.betas[0], self.betas[1], self.eps, self.wd,
                                  self.step_count, stream()), "adamw")
        # the kernel wrote through raw pointers (no autograd version bump): invalidate the packed-weight cache
        mark_weights_dirty(self.model)

    def step(self, img, targets, lengths, keep_mask=None, lr=None):
        loss = self.forward_backward(img, targets, lengths, keep_mask)
        self.optimizer_step(lr)
        return loss

    def capture_step(self, img, max_target_len, masked=True, single_stream=True):
        """Record step() -- gradient clear, forward, CTC, backward (incl. the bucketed all-reduces when a process group
        is in use), AdamW -- into ONE HIP graph (~360 launches per step are launch-bound at small per-rank
        batches).  Call after at least one eager step() of the same batch shape (lazily sized workspaces, one-time
        kernel attributes and the engine's streams exist then).  Returns a GraphedStep; its step() has step()'s
        signature and results (bit-identical: same kernels, same order, same scalars)."""
        return GraphedStep(self, img, max_target_len, masked, single_stream)

    def sam_first_step(self, rho=0.05):
        """SAM.first_step (utils/sam.py:15-27, adaptive=False) on the flat buffers: |g| by a two-stage reproducible sum,
        old_w = w, w += rho g / (|g| + 1e-12).  Two launches."""
        from ._lib import check, lib
        from .ops import ptr, stream
        fl = self.flat
        n = fl.flat_p.numel()
        if not hasattr(self, "_sam_buf"):
            self._sam_buf = (torch.empty_like(fl.flat_p), torch.empty(lib.htrvt_sumsq_blocks(n), device=fl.flat_p.device),
                             torch.empty(1, device=fl.flat_p.device))
        old_p, partial, norm_sq = self._sam_buf
        check(lib.htrvt_sumsq(ptr(fl.flat_g), n, ptr(partial), ptr(norm_sq), stream()), "sumsq")
        check(lib.htrvt_sam_first_step(ptr(fl.flat_p), ptr(fl.flat_g), ptr(old_p), n, float(rho), ptr(norm_sq), stream()),
              "sam_first_step")
        mark_weights_dirty(self.model)

    def sam_second_step(self, lr=None):
        """SAM.second_step (utils/sam.py:29-38): w = old_w, then the base optimizer (AdamW) with the gradients taken at
        the perturbed point."""
        from ._lib import check, lib
        from .ops import ptr, stream
        fl = self.flat
        check(lib.htrvt_sam_restore(ptr(fl.flat_p), ptr(self._sam_buf[0]), fl.flat_p.numel(), stream()), "sam_restore")
        self.optimizer_step(lr)

    def sam_step(self, img, targets, lengths, keep_mask=None, keep_mask2=None, lr=None, rho=0.05):
        """One iteration of the reference's optimizer, SAM(AdamW) (train.py:119-126, utils/sam.py:15-38, adaptive=False):
        gradients at w -> climb to w + rho g/|g| -> gradients there (a second, independently masked pass) -> back to w
        -> AdamW with the second gradients.  Three flat launches around the two forward/backward passes; under data
        parallelism both passes all-reduce, so every rank computes the same norm.  Returns the first-pass loss."""
        loss = self.forward_backward(img, targets, lengths, keep_mask)
        self.sam_first_step(rho)
        self.forward_backward(img, targets, lengths, keep_mask2)
        self.sam_second_step(lr)
        return loss


class GraphedStep:
    """Trainer.step as a replayed HIP graph.

    What a graph bakes and what changes per step decides the layout: every per-step INPUT lives in a fixed device buffer
    that the host refreshes in stream order before the replay -- the image batch (`img`; fill it in place or pass a
    tensor to step(), copied device-to-device), the label arrays (`tg`, `tl`, `off`, sized for `max_target_len` labels
    per line: the CTC kernels are launched for that bound), the span keep-mask (`keep`), and AdamW's derived scalars
    (`hyper`: lr and the step number change every iteration, htrvt_adamw_dev reads them from device memory).  Host
    staging goes through pinned buffers; before re-filling them the host waits for the previous step's copies, which
    also bounds its run-ahead to one step (what Engine.forward's throttle does for the eager path)."""

    def __init__(self, tr, img, max_target_len, masked=True, single_stream=True):
        import numpy as np
        from ._lib import lib, HTRVT_ADAMW_SCALARS
        self.tr, self.np = tr, np
        eng, fl = tr.engine, tr.flat
        if tr.step_count < 1:
            raise RuntimeError("capture_step: run one eager step() of this batch shape first")
        dev = img.device
        B, N = img.shape[0], tr.model.num_patches
        self.B, self.N, self.maxlen = B, N, int(max_target_len)
        assert 1 <= self.maxlen
        cap = B * self.maxlen
        self.img = img.detach().clone().contiguous()
        self.keep = torch.ones(N, dtype=torch.float32, device=dev) if masked else None
        self.tg = torch.zeros(cap, dtype=torch.int32, device=dev)
        self.tl = torch.zeros(B, dtype=torch.int32, device=dev)
        self.off = torch.zeros(B, dtype=torch.int32, device=dev)
        self.hyper = torch.zeros(HTRVT_ADAMW_SCALARS, dtype=torch.float32, device=dev)
        self._h_tg = torch.zeros(cap, dtype=torch.int32).pin_memory()
        self._h_tl = torch.zeros(B, dtype=torch.int32).pin_memory()
        self._h_off = torch.zeros(B, dtype=torch.int32).pin_memory()
        self._h_keep = torch.ones(N, dtype=torch.float32).pin_memory()
        self._h_hyper = torch.zeros(HTRVT_ADAMW_SCALARS, dtype=torch.float32).pin_memory()
        self._staged = None
        # the recorded launches bake the engine's execution plan: its boolean switches (and the override environment that
        # sets them) must be the same at every replay
        self._plan = self._engine_plan(eng)
        fl.check_views()
        mark_weights_dirty(tr.model)       # the recorded forward must contain the weight re-layout launch
        self.graph = torch.cuda.CUDAGraph()
        eng.capturing = True
        saved_single, eng.single_stream = eng.single_stream, bool(single_stream)
        try:
            # thread_local: the process group's watchdog thread polls its events while we capture; under the default (global)
            # mode such a call from ANOTHER thread invalidates the capture (seen as an abort in the one-rank RCCL test)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss = tr._forward_backward_staged(self.img, (self.tg, self.tl, self.off, self.maxlen), self.keep)
                from ._lib import check
                from .ops import ptr, stream
                check(lib.htrvt_adamw_dev(ptr(fl.flat_p), ptr(fl.flat_g), ptr(tr.flat_m), ptr(tr.flat_v), fl.flat_p.numel(),
                                          ptr(self.hyper), stream()), "adamw_dev")
        except BaseException:
            # a refused capture leaves the engine mid-step: drop what it kept so that the eager path works again
            eng._side_active, eng.saved, eng._zoff, eng._pending_unpack = False, None, None, []
            raise
        finally:
            eng.capturing = False
            eng.single_stream = saved_single
        mark_weights_dirty(tr.model)

    @staticmethod
    def _engine_plan(eng):
        import os
        return (tuple(sorted((k, v) for k, v in vars(eng).items() if isinstance(v, bool) and k not in ("capturing", "single_stream", "_side_active", "_bn_train", "_saving"))),
                os.environ.get("HTRVT_ENGINE_OVERRIDE", ""))

    def step(self, img, targets, lengths, keep_mask=None, lr=None):
        """one replay; returns the (device, reused) mean-loss tensor of this step"""
        from ._lib import lib
        np, tr = self.np, self.tr
        tr.flat.check_views()
        if self._engine_plan(tr.engine) != self._plan:
            raise RuntimeError("GraphedStep: the engine's switches changed since the capture (the graph holds the launches of the "
                               "plan it was recorded with): capture the step again")
        if self._staged is not None:
            self._staged.synchronize()      # previous step's host->device copies have run: the pinned buffers are free
        tl = np.asarray(lengths, dtype=np.int32).reshape(-1)
        tg = np.asarray(targets, dtype=np.int32).reshape(-1)
        if tl.shape[0] != self.B or (tl.size and int(tl.max()) > self.maxlen) or tg.shape[0] > self._h_tg.numel():
            raise ValueError(f"GraphedStep was captured for {self.B} lines of <= {self.maxlen} labels")
        if tg.shape[0] != int(tl.sum()):     # the offsets below index the label buffer: a short `targets` would leave an earlier step's labels in reach
            raise ValueError(f"GraphedStep: {tg.shape[0]} labels for lengths that sum to {int(tl.sum())}")
        off = np.zeros_like(tl)
        if tl.size > 1:
            off[1:] = np.cumsum(tl[:-1])
        self._h_tl.numpy()[:] = tl
        self._h_off.numpy()[:] = off
        self._h_tg.numpy()[:tg.shape[0]] = tg
        for h, d in ((self._h_tg, self.tg), (self._h_tl, self.tl), (self._h_off, self.off)):
            d.copy_(h, non_blocking=True)
        if (keep_mask is None) != (self.keep is None):
            raise ValueError("GraphedStep: captured with" + ("" if self.keep is not None else "out") + " a span mask")
        if keep_mask is not None:
            if keep_mask.is_cuda:
                self.keep.copy_(keep_mask, non_blocking=True)
            else:
                self._h_keep.copy_(keep_mask.to(torch.float32).reshape(-1))
                self.keep.copy_(self._h_keep, non_blocking=True)
        if img is not self.img and img.data_ptr() != self.img.data_ptr():
            self.img.copy_(img, non_blocking=True)
        tr.step_count += 1
        hy = self._h_hyper.numpy()
        import ctypes
        lib.htrvt_adamw_scalars(float(tr.lr if lr is None else lr), tr.betas[0], tr.betas[1], tr.eps, tr.wd, tr.step_count,
                                ctypes.c_void_p(hy.ctypes.data))
        self.hyper.copy_(self._h_hyper, non_blocking=True)
        self._staged = torch.cuda.Event()
        self._staged.record()
        with trace.range_("htrvt.graph_step"):
            self.graph.replay()
        mark_weights_dirty(tr.model)
        return self.loss
