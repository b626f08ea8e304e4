"""Exponential moving average of a model's state_dict -- the reference's `utils.ModelEma`
(/root/reference/model_v1/utils/utils.py:128-173) with `update` as ONE multi-tensor HIP launch
(htrvt_ema_update) instead of ~150 tiny elementwise launches per iteration.

Same constructor / `update(model, num_updates)` / `.ema` surface; the decay warm-up
`min(decay, (1 + n) / (10 + n))` and the treatment of every state_dict entry (BatchNorm running statistics and
the int64 `num_batches_tracked` counters included: float math, truncating store) follow utils.py:158-173."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from copy import deepcopy

import torch


class _Entry(C.Structure):      # include/htrvt.h: HtrvtEmaEntry
    _fields_ = [("ema", C.c_void_p), ("model", C.c_void_p), ("numel", C.c_int64), ("is_int64", C.c_int32), ("pad_", C.c_int32)]


class ModelEma:
    def __init__(self, model, decay=0.9999, device="", resume=""):
        self.ema = deepcopy(model)
        self.ema.eval()
        self.decay = decay
        self.device = device
        if device:
            self.ema.to(device=device)
        self.ema_has_module = hasattr(self.ema, "module")
        if resume:
            self._load_checkpoint(resume)
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._table = None      # (key, device table tensor, count, max_numel)

    def _load_checkpoint(self, checkpoint_path, mapl=None):
        checkpoint = torch.load(checkpoint_path, map_location=mapl)
        assert isinstance(checkpoint, dict)
        if "state_dict_ema" not in checkpoint:
            print("=> Failed to find state_dict_ema, starting from loaded model weights")
            return
        renamed = OrderedDict()
        for k, v in checkpoint["state_dict_ema"].items():
            renamed["module." + k if self.ema_has_module and not k.startswith("module") else k] = v
        self.ema.load_state_dict(renamed)
        print("=> Loaded state_dict_ema")

    def _entries(self, model):
        needs_module = hasattr(model, "module") and not self.ema_has_module
        msd = model.state_dict()
        pairs = []
        for k, ema_v in self.ema.state_dict().items():
            model_v = msd["module." + k if needs_module else k]
            pairs.append((ema_v, model_v))
        return pairs

    def update(self, model, num_updates=-1):
        from ._lib import check, lib
        from .ops import stream
        cdecay = min(self.decay, (1 + num_updates) / (10 + num_updates)) if num_updates >= 0 else self.decay
        pairs = self._entries(model)
        key = tuple((e.data_ptr(), m.data_ptr()) for e, m in pairs)
        if self._table is None or self._table[0] != key:
            arr = (_Entry * len(pairs))()
            for i, (e, m) in enumerate(pairs):
                if not (e.is_cuda and m.is_cuda and e.device == m.device):
                    raise RuntimeError("htrvt_amd.ModelEma needs the model and its average on one MI355X (no CPU fallback)")
                if e.dtype != m.dtype or e.numel() != m.numel() or e.dtype not in (torch.float32, torch.int64):
                    raise RuntimeError(f"ModelEma: unsupported state_dict entry ({e.dtype}, {tuple(e.shape)})")
                if not (e.is_contiguous() and m.is_contiguous()):
                    raise RuntimeError("ModelEma: state_dict entries must be contiguous")
                arr[i] = _Entry(e.data_ptr(), m.data_ptr(), e.numel(), 1 if e.dtype == torch.int64 else 0, 0)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self._table = (key, host.to(pairs[0][0].device), len(pairs), max(e.numel() for e, _ in pairs))
        _, table, count, max_numel = self._table
        with torch.no_grad():
            check(lib.htrvt_ema_update(table.data_ptr(), count, max_numel, float(cdecay), stream()), "ema_update")
        # the launch wrote the averaged weights through raw pointers: drop the engines' packed / cast copies of them
        from . import mark_weights_dirty
        mark_weights_dirty(self.ema.module if self.ema_has_module else self.ema)
