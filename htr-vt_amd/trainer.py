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

from . import mark_weights_dirty


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
        from .ctc import ctc_forward_backward, stage_targets
        eng, fl = self.engine, self.flat
        fl.check_views()     # a caller that re-bound p.data (reference SAM.second_step, model.to()) must rebuild the Trainer
        staged = stage_targets(targets, lengths, img.device)   # before the forward is enqueued (see stage_targets)
        fl.flat_g.zero_()
        y = eng.forward(self.P, img, keep_mask=keep_mask, train=True, save=True)
        nll, dy = ctc_forward_backward(y, targets, lengths, want_grad=True, grad_scale=1.0 / self.world, staged=staged)
        eng.backward(self.P, fl.G, dy, after_encoder=fl.reduce_encoder_bucket, after_layer3=fl.reduce_layer3_bucket)
        fl.reduce_stem_bucket()
        return nll.mean()

    def optimizer_step(self, lr=None):
        from ._lib import check, lib
        from .ops import ptr, stream
        self.step_count += 1
        fl = self.flat
        check(lib.htrvt_adamw(ptr(fl.flat_p), ptr(fl.flat_g), ptr(self.flat_m), ptr(self.flat_v), fl.flat_p.numel(),
                              float(self.lr if lr is None else lr), self.betas[0], self.betas[1], self.eps, self.wd,
                              self.step_count, stream()), "adamw")
        # the kernel wrote through raw pointers (no autograd version bump): invalidate the packed-weight cache
        mark_weights_dirty(self.model)

    def step(self, img, targets, lengths, keep_mask=None, lr=None):
        loss = self.forward_backward(img, targets, lengths, keep_mask)
        self.optimizer_step(lr)
        return loss

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
