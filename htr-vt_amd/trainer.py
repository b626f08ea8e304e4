"""Data-parallel training step of the HTR-VT hot path (one process per MI355X).

step = forward + fused CTC + backward (+ gradient all-reduce over RCCL/xGMI)
+ AdamW, i.e. the inner statements of /root/reference/model_v1/train.py:119-126
without the SAM second pass (SURVEY.md 8(f-1) lists SAM as a later row).

MI355X-first layout: every trainable parameter is a view into ONE flat float32
buffer (same for gradients and the two Adam moments), so the optimizer is one
kernel launch and the gradient exchange is two large all-reduces (xGMI is
point-to-point: few, large messages).  The encoder/head bucket -- the tail of
the flat buffer, complete early in backward -- is reduced on a side stream while
the stem backward (~80 % of the FLOPs) still runs; the stem bucket follows.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from ._lib import check, lib
from .ctc import ctc_forward_backward
from .ops import ptr, stream


class Trainer:
    def __init__(self, model, max_lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.5, world_size=1):
        self.model = model
        self.world = world_size
        self.lr, self.betas, self.eps, self.wd = max_lr, betas, eps, weight_decay
        self.step_count = 0
        dev = next(model.parameters()).device
        assert dev.type == "cuda", "Trainer needs the model on an MI355X (model.cuda())"
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        total = sum((p.numel() + 3) // 4 * 4 for _, p in named)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.G = {}
        off = 0
        self.enc_start = None
        for n, p in named:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p)
            self.G[n] = self.flat_g[off:off + k].view_as(p)
            if self.enc_start is None and n.startswith("blocks."):
                self.enc_start = off
            off += (k + 3) // 4 * 4
        if self.enc_start is None:
            self.enc_start = 0
        self.P = dict(model.state_dict(keep_vars=True))
        self.engine = model._engine(dev)
        self.side = torch.cuda.Stream(device=dev) if world_size > 1 else None
        if world_size > 1:   # identical replicas: broadcast rank 0's parameters and BN buffers
            dist.broadcast(self.flat_p, 0)
            for n, b in model.named_buffers():
                dist.broadcast(b, 0)

    def _check_views(self):
        for n, p in self.model.named_parameters():
            if p.requires_grad and p.data.untyped_storage().data_ptr() != self.flat_p.untyped_storage().data_ptr():
                raise RuntimeError(f"parameter {n} was re-bound outside the flat buffer; rebuild the Trainer")

    def forward_backward(self, img, targets, lengths, keep_mask=None):
        """one forward + CTC + backward on this rank's shard; gradients (already divided by world_size)
        land in the flat gradient buffer.  Returns the local mean loss (device scalar)."""
        eng = self.engine
        self.flat_g.zero_()
        y = eng.forward(self.P, img, keep_mask=keep_mask, train=True, save=True)
        nll, dy = ctc_forward_backward(y, targets, lengths, want_grad=True, grad_scale=1.0 / self.world)
        if self.world > 1:
            eng.backward(self.P, self.G, dy, after_encoder=self._reduce_encoder_bucket)
            dist.all_reduce(self.flat_g[:self.enc_start])           # stem bucket, current stream
            torch.cuda.current_stream().wait_stream(self.side)
        else:
            eng.backward(self.P, self.G, dy)
        return nll.mean()

    def _reduce_encoder_bucket(self):
        """called by Engine.backward once every blocks.*/norm/head gradient is enqueued"""
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            dist.all_reduce(self.flat_g[self.enc_start:])

    def optimizer_step(self, lr=None):
        self.step_count += 1
        check(lib.htrvt_adamw(ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), self.flat_p.numel(),
                              float(self.lr if lr is None else lr), self.betas[0], self.betas[1], self.eps, self.wd,
                              self.step_count, stream()), "adamw")
        # the kernel wrote through raw pointers (no autograd version bump): invalidate the packed-weight cache
        self.engine.weights_epoch += 1

    def step(self, img, targets, lengths, keep_mask=None, lr=None):
        loss = self.forward_backward(img, targets, lengths, keep_mask)
        self.optimizer_step(lr)
        return loss
