"""Fused log-softmax + CTC loss (htrvt_ctc_loss) as an autograd function.

Counterpart of `compute_loss` in /root/reference/model_v1/train.py:21-30:
    preds.permute(1,0,2).log_softmax(2); CTCLoss(reduction='none', zero_infinity=True)(...).mean()
with input length N for every sample and blank = 0."""
import torch

from ._lib import check, lib
from .ops import ptr, stream


def _prep_targets(targets, target_lengths, device):
    tl = torch.as_tensor(target_lengths, dtype=torch.int32).cpu()
    off = torch.zeros_like(tl)
    if tl.numel() > 1:
        off[1:] = torch.cumsum(tl[:-1], 0)
    maxlen = int(tl.max()) if tl.numel() else 0
    tg = torch.as_tensor(targets, dtype=torch.int32)
    if tg.numel() == 0:
        tg = torch.zeros(1, dtype=torch.int32)
    return _upload(tg, device), _upload(tl, device), _upload(off, device), maxlen


def _upload(t, device):
    """host -> device through pinned memory, stream-ordered: a copy from pageable memory would block the host until
    everything already enqueued (the previous training step) has finished, once per step"""
    if t.is_cuda or torch.device(device).type != "cuda":
        return t.to(device)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def stage_targets(targets, target_lengths, device):
    """upload the label arrays (three small host->device copies through pinned memory: the host does not wait for the
    device)."""
    return _prep_targets(targets, target_lengths, device)


def ctc_forward_backward(logits, targets, target_lengths, want_grad=True, grad_scale=1.0, staged=None):
    """logits [B,T,C] float32 (device).  Returns (nll [B], dmean/dlogits [B,T,C] or None)."""
    assert logits.is_cuda and logits.dtype == torch.float32
    logits = logits.contiguous()
    B, T, C = logits.shape
    tg, tl, off, maxlen = staged if staged is not None else _prep_targets(targets, target_lengths, logits.device)
    nll = torch.empty(B, dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits) if want_grad else None
    ws = torch.empty(lib.htrvt_ctc_workspace_floats(B, T, maxlen), dtype=torch.float32, device=logits.device)
    check(lib.htrvt_ctc_loss(ptr(logits), ptr(tg), ptr(tl), ptr(off), ptr(nll), ptr(grad), ptr(ws), B, T, C, maxlen,
                             float(grad_scale), stream()), "ctc_loss")
    return nll, grad


class _CTCMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, target_lengths):
        nll, grad = ctc_forward_backward(logits, targets, target_lengths, want_grad=True)
        ctx.save_for_backward(grad)
        return nll.mean()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def ctc_loss(logits, targets, target_lengths):
    """mean over the batch of the per-sample CTC negative log-likelihood (not length-normalised)."""
    return _CTCMean.apply(logits, targets, target_lengths)


def greedy_decode(logits, ncharacter=None):
    """valid.py:40-42 + CTCLabelConverter.decode (utils/utils.py:72-86) on the device: per frame arg-max of the
    logits (= arg-max of their log-softmax), blanks / repeats / indices >= ncharacter dropped.
    logits [B,T,C] float32 -> (idx [B,T] int32 left-packed, lens [B] int32); ncharacter = len(converter.character)."""
    from ._lib import check, lib
    from .ops import ptr, stream
    if not logits.is_cuda:
        raise RuntimeError("htrvt_amd.greedy_decode needs a device tensor on an MI355X (no CPU fallback)")
    logits = logits.float().contiguous()
    B, T, Cc = logits.shape
    out = torch.zeros(B, T, dtype=torch.int32, device=logits.device)
    lens = torch.empty(B, dtype=torch.int32, device=logits.device)
    check(lib.htrvt_ctc_greedy_decode(ptr(logits), B, T, Cc, Cc, int(Cc if ncharacter is None else ncharacter), ptr(out),
                                      ptr(lens), stream()), "ctc_greedy_decode")
    return out, lens
