"""Host-side execution plan of the HTR-VT hot path on one MI355X.

`Engine.forward` / `Engine.backward` enqueue the gfx950 kernels of
libhtrvt_hip.so in the order of the reference forward
(/root/reference/model_v1/model/HTR_VT.py:222-241, resnet18.py:73-84) and of its
autograd backward (train.py:123).  All arithmetic is in the HIP kernels; torch
only allocates buffers (`torch.empty/zeros`) and provides the stream.

Activation layout: NHWC for the stem, [B,N,D] for tokens, element type
`dtype` (float32 = parity path on f32 MFMA, bfloat16 = throughput path);
statistics, logits, parameters and parameter gradients are float32.
"""
from __future__ import annotations

import math

import os

import torch

from . import ops
from ._lib import lib, check, RelayoutJob, RELAYOUT_PACK_CONV, RELAYOUT_CAST_TRANSPOSE, RELAYOUT_UNPACK_WGRAD, RELAYOUT_ARG_JOBS
from .ops import KMAJOR, MNMAJOR, GATHER_CONV_DGRAD, GATHER_CONV_FWD, GATHER_CONV_WGRAD, ConvGeom, cpad, dt, gemm, ptr, stream

LN_EPS = 1e-6       # HTR_VT.py:252
WHITEN_EPS = 1e-5   # HTR_VT.py:136
BN_EPS = 1e-5       # resnet18.py:16
BN_MOMENTUM = 0.1


class ModelShape:
    """Static shape of one model (mirrors MaskedAutoencoderViT.__init__, HTR_VT.py:143-172)."""

    def __init__(self, nb_cls, img_size, embed_dim, depth, num_heads, mlp_ratio=4.0, patch_size=(4, 64), ln_eps=LN_EPS):
        self.nb_cls = int(nb_cls)
        self.ln_eps = float(ln_eps)
        self.H, self.W = int(img_size[0]), int(img_size[1])
        self.D, self.depth, self.heads = int(embed_dim), int(depth), int(num_heads)
        self.hd = self.D // self.heads
        self.hidden = int(embed_dim * mlp_ratio)
        self.grid = (self.H // patch_size[0], self.W // patch_size[1])
        self.num_patches = self.grid[0] * self.grid[1]
        assert self.D % 32 == 0 and self.D % self.heads == 0
        assert self.H % 64 == 0 and self.W % 64 == 0, "img_size must be a multiple of 64 (HTR_VT.py:158-160)"

    def stem_convs(self):
        """(param prefix, Ci, Co, k, stride, pad) of every MFMA conv in execution order (resnet18.py:52-71)."""
        D = self.D
        out, inpl = [], D // 4
        for li, (planes, stride) in enumerate(((D // 4, (2, 1)), (D // 2, (2, 2)), (D, (2, 2))), start=1):
            p = f"patch_embed.layer{li}"
            out.append((f"{p}.0.conv1", inpl, planes, 3, stride, 1))
            out.append((f"{p}.0.conv2", planes, planes, 3, (1, 1), 1))
            out.append((f"{p}.0.downsample.0", inpl, planes, 1, stride, 0))
            out.append((f"{p}.1.conv1", planes, planes, 3, (1, 1), 1))
            out.append((f"{p}.1.conv2", planes, planes, 3, (1, 1), 1))
            inpl = planes
        return out

    def linears(self):
        names = []
        for i in range(self.depth):
            names += [f"blocks.{i}.attn.qkv", f"blocks.{i}.attn.proj", f"blocks.{i}.mlp.fc1", f"blocks.{i}.mlp.fc2"]
        return names + ["head"]


def _job(kind, src, dst0, dst1, d0, d1, taps=0, cpad_in=0, cpad_out=0, row_taps=0, tap0=0):
    j = RelayoutJob()
    j.src, j.dst0, j.dst1, j.kind, j.d0, j.d1 = ptr(src), ptr(dst0), ptr(dst1), kind, d0, d1
    j.taps, j.cpad_in, j.cpad_out, j.row_taps, j.tap0 = taps, cpad_in, cpad_out, row_taps, tap0
    return j


class Engine:
    def __init__(self, shape: ModelShape, dtype=torch.float32, device="cuda", split_bf16=False):
        self.s = shape
        self.dtype = dtype
        self.dev = torch.device(device)
        self.dti = dt(dtype)
        # Split-bfloat16 parity path (csrc/split.hip): activations, statistics and every non-GEMM kernel stay float32 exactly
        # as on the float32 path; the convolutions and the encoder's Linear layers -- 98 % of the FLOPs -- run on the bf16
        # matrix cores with operands split into hi + lo bf16 parts concatenated along K (three products, float32
        # accumulate).  `gdt` is the element type the big GEMMs' operands have.
        self.split = bool(split_bf16)
        assert not self.split or dtype == torch.float32, "split_bf16 is a mode of the float32 path"
        self.gdt = torch.bfloat16 if self.split else dtype
        self._split_cache = []       # backward: the few most recent (source tensor, cat, hi, lo) splits (a gradient feeds dgrad AND wgrad)
        self._saved_planes, self._saving = {}, False    # forward(save=True): id(activation) -> its hi / lo planes, for the weight gradients
        self._packs = {}      # name -> (version key, tensors)
        self.weights_epoch = 0   # bumped by whoever rewrites parameters through raw pointers (Trainer.optimizer_step)
        self.fuse_conv1_backward = True   # conv1/bn1/maxpool backward as per-channel sums over the pooled gradient
        self.fuse_bn_backward = True   # bf16: ReLU mask + BN-backward sums in the dgrad epilogue (False: separate pass)
        self.overlap_wgrad = True      # weight-gradient GEMMs on a side stream
        # conv1 + BN + ReLU + max-pool in one pass over the image (no conv1 tensor): the throughput path.  The float32 parity
        # path keeps the two-kernel form by default -- its batch statistics are then summed over the conv outputs the way
        # the reference sums them, which keeps the sign-level comparison of Adam's first updates against the reference
        # trace (tests/test_train_iter_gpu.py) where it was; tests/test_stem_gpu.py runs the fused form in float32 too.
        self.fuse_stem_forward = dtype == torch.bfloat16
        self.fused_attention = True    # bf16: one launch per direction, scores / probabilities never reach HBM (csrc/attention.hip)
        # strided conv dgrad = one launch per input-pixel parity class: independent launches (disjoint output pixels), so
        # they may run on separate streams -- an epilogue-only class (no tap reaches it) then overlaps an MFMA-heavy one
        self.parallel_classes = False
        self._cls_streams = None
        # training steps rewrite every weight: re-pack / re-cast all of them on the side stream at the start of the forward,
        # under the (HBM-bound) image statistics, conv1 and max-pool kernels, instead of ~30 latency-bound launches in
        # front of their first use
        self.prefetch_packs = True
        # bf16: every conv / Linear weight is re-packed by ONE table-driven launch per step and the conv weight gradients are
        # unpacked by one launch per DP bucket (csrc/relayout.hip) instead of one launch per tensor (47 per step)
        self.table_relayout = True
        self.relu_bitmask = True        # a block's output ReLU: the forward BatchNorm pass writes its 1-bit mask, the fused dgrad reads that instead of the activation
        self.relu_mask_from_bn = True   # conv2's fused dgrad epilogue: ReLU mask from the BatchNorm input it reads anyway (no read of a1)
        self.merge_bn_backward = True   # first block of a stage: bn2 + downsample-BN backward in one pass over the shared gradient
        self._pending_unpack = []
        # first block of a stage, bf16: the input gradient of the 1x1 downsample conv is formed INSIDE the class-(0,0) launch
        # of the strided 3x3 conv's dgrad (one more tap, HtrvtGemmDesc.A2) instead of by its own parity-class launches
        # plus a residual round trip of the whole input gradient
        self.fuse_downsample_dgrad = True
        self.halo_wgrad = True         # 3x3 stride-1 conv weight gradients on the halo-staged kernel (csrc/gemm_hwgrad_impl.h)
        # strided 3x3 conv dgrad: all parity classes in ONE launch on halo-staged tiles (csrc/gemm_halo_impl.h,
        # gemm_halo_s2_kernel) instead of one gather launch per class
        self.merged_strided_dgrad = True
        # split-K weight gradients through per-K-range slabs + an ordered sum instead of float atomics: bitwise reproducible
        # run to run (tests/test_determinism_gpu.py).  Round 3: also the bf16 default -- equal to the atomic form at 64-128
        # images per GPU (36.39 vs 36.41 ms), faster below (B = 32: 11.33 vs 11.48 ms, B = 16: 7.22 vs 7.57 ms: float atomics
        # run at ~1.3 TB/s chip-wide, slab stores + the ordered sum at HBM speed)
        self.deterministic = True
        self._side, self._side_active = None, False
        self._call_started = None      # event at the start of the previous forward() (host run-ahead throttle)
        self.capturing = False         # True while Trainer.capture_step records the step into a HIP graph
        # every launch of a step on ONE stream (no weight-gradient / weight-pack side streams): what a captured graph wants
        # on this runtime -- hipGraphLaunch resolves cross-stream edges of a multi-stream capture node by node (measured: a
        # graph of the four-stream step replays SLOWER than the eager launches, 14.3 vs 6.9 ms at 16 images)
        self.single_stream = False
        self.saved = None
        self._bn_train = True
        self._zarena, self._zoff, self._zneed, self._zneed_max = None, None, 0, 0
        # A/B runs on one box: HTRVT_ENGINE_OVERRIDE="relu_mask_from_bn=0,table_relayout=0" flips boolean switches above
        for kv in filter(None, os.environ.get("HTRVT_ENGINE_OVERRIDE", "").split(",")):
            k, _, v = kv.partition("=")
            if not isinstance(getattr(self, k.strip(), None), bool):
                raise ValueError(f"HTRVT_ENGINE_OVERRIDE: no boolean engine switch {k!r}")
            setattr(self, k.strip(), v.strip() not in ("0", "false", "False", ""))

    # ------------------------------------------------------------------ small helpers
    def _empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.dev)

    def _zeros(self, *shape, dtype=torch.float32):
        """float32 zeros for atomically accumulated outputs.  Inside backward() they are slices of ONE arena cleared by
        one memset per step (31 fill launches otherwise, each with its ~5 us dispatch gap)."""
        if dtype is not torch.float32 or self._zoff is None:
            return torch.zeros(*shape, dtype=dtype, device=self.dev)
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) // 64 * 64
        self._zneed += n_al
        if self._zarena is None or self._zoff + n_al > self._zarena.numel():
            return torch.zeros(*shape, dtype=dtype, device=self.dev)
        t = self._zarena[self._zoff:self._zoff + n].view(*shape)
        self._zoff += n_al
        return t

    def _zarena_begin(self):
        if self._zarena is None or self._zarena.numel() < self._zneed_max:
            self._zarena = torch.zeros(max(self._zneed_max, 1), dtype=torch.float32, device=self.dev) if self._zneed_max else None
        elif self._zarena is not None:
            self._zarena.zero_()
        self._zoff, self._zneed = 0, 0

    def _zarena_end(self):
        self._zneed_max = max(self._zneed_max, self._zneed)
        self._zoff = None

    def _wkey(self, t):
        return (t.data_ptr(), t._version, self.weights_epoch)

    # ------------------------------------------------------------------ split-bf16 operands (csrc/split.hip)
    def _split(self, src, rows, cols, order=0, cat=True, planes=False, cat_f32=False, transpose=False):
        """float32 [rows][cols] (contiguous) -> (cat [rows][3 cols] | transposed [cols][3 rows], hi, lo)"""
        BF = torch.bfloat16
        c = None
        if cat:
            shp = (cols, 3 * rows) if transpose else (rows, 3 * cols)
            c = torch.empty(*shp, dtype=torch.float32 if cat_f32 else BF, device=self.dev)
        hi = torch.empty(rows, cols, dtype=BF, device=self.dev) if planes else None
        lo = torch.empty(rows, cols, dtype=BF, device=self.dev) if planes else None
        check(lib.htrvt_split_bf16(ptr(src), rows, cols, cols, ptr(c), order, 1 if cat_f32 else 0, 1 if transpose else 0,
                                   ptr(hi), ptr(lo), stream()), "split_bf16")
        return c, hi, lo

    @staticmethod
    def _batch_slices(B, nbytes, rows_per_image):
        """smallest number of equal batch slices that keeps an operand of `nbytes` below 2 GiB per launch (M tiles of 256
        rows must not straddle slices: the per-tile column sums are concatenated)"""
        def fits(n):
            return B % n == 0 and nbytes // n < 2 ** 31 - 4096 and (n == 1 or (B // n * rows_per_image) % 256 == 0)
        n = 1
        while n < B and not fits(n):
            n += 1
        if not fits(n):
            raise ValueError(f"no batch slicing of {B} images keeps a {nbytes}-byte operand below 2 GiB per launch with whole 256-row tiles "
                             f"({rows_per_image} rows per image)")
        return n

    def _split_act(self, t, cols, cat=True, planes=False):
        """split of an activation / gradient tensor whose innermost extent is `cols`, remembered while the backward may ask
        for it again (the same gradient is the A operand of a dgrad launch and the B operand of three wgrad launches)"""
        cur = torch.cuda.current_stream()
        kept = self._saved_planes.get(id(t))      # a forward that saves for backward split this activation into planes as well
        for ent in self._split_cache + ([kept] if kept is not None else []):
            if ent[0] is t and (ent[1] is not None or not cat) and (ent[2] is not None or not planes):
                if ent[4] != cur:      # made on the main stream, read by a weight-gradient launch on the side stream: the
                    for u in ent[1:4]:  # caching allocator must not hand the block out again before that launch has run
                        if u is not None:
                            u.record_stream(cur)
                return ent[1], ent[2], ent[3]
        want_planes = planes or self._saving      # forward of a training step: the weight gradient will want the planes of this input
        c, hi, lo = self._split(t, t.numel() // cols, cols, order=0, cat=cat, planes=want_planes)
        self._split_cache = [e for e in self._split_cache if e[0] is not t][-2:] + [(t, c, hi, lo, cur)]
        if self._saving:
            self._saved_planes[id(t)] = (t, None, hi, lo, cur)
        return c, hi, lo

    def _lin_w_split(self, name, w):
        """([out][3 in] (hi | hi | lo) for the forward, [in][3 out] of the transposed weight for the dgrad)"""
        key = self._wkey(w)
        ent = self._packs.get(name + "/split")
        if ent is None or ent[0] != key:
            out_f, in_f = w.shape
            ws, _, _ = self._split(w, out_f, in_f, order=1)
            wts, _, _ = self._split(w, out_f, in_f, order=1, transpose=True)
            self._packs[name + "/split"] = (key, (ws, wts))
            return ws, wts
        return ent[1]

    def _conv_w_split(self, name, w):
        """forward pack [Co][taps][Cpad(3 Ci)] and dgrad pack [Ci][taps][Cpad(3 Co)] of the split weight: the conv as seen by
        the bf16 kernels has 3 Ci input channels (hi | hi | lo copies of the weight against (hi | lo | hi) activations)"""
        key = self._wkey(w)
        ent = self._packs.get(name + "/split")
        if ent is None or ent[0] != key:
            Co, Ci, kh, kw = w.shape
            taps = kh * kw
            BF = torch.bfloat16
            cpi3, cpo3 = cpad(3 * Ci, BF), cpad(3 * Co, BF)
            fwd = torch.zeros(Co, taps, cpi3, dtype=BF, device=self.dev)
            dgr = torch.zeros(Ci, taps, cpo3, dtype=BF, device=self.dev)
            v, _, _ = self._split(w, Co, Ci * taps, order=1, cat_f32=True)          # [Co][3][Ci][taps] = a [Co, 3 Ci, kh, kw] weight
            check(lib.htrvt_pack_conv_weight(ptr(v), ptr(fwd), None, Co, 3 * Ci, taps, cpi3, cpad(Co, BF), dt(BF), stream()), "pack_conv_weight")
            v, _, _ = self._split(w, 1, Co * Ci * taps, order=1, cat_f32=True)      # [3][Co][Ci][taps] = a [3 Co, Ci, kh, kw] weight
            check(lib.htrvt_pack_conv_weight(ptr(v), None, ptr(dgr), 3 * Co, Ci, taps, cpad(Ci, BF), cpo3, dt(BF), stream()), "pack_conv_weight")
            self._packs[name + "/split"] = (key, (fwd, dgr))
            return fwd, dgr
        return ent[1]

    def _lin_w(self, name, w):
        """(weight, transposed weight) of a Linear in compute dtype: [out,in] for the forward GEMM and [in,out] for the
        dgrad GEMM dx = dy @ w, so that both are K-major x K-major products on the same kernel.  float32: (w, None)."""
        if self.split:
            return self._lin_w_split(name, w)
        if self.dtype == torch.float32:
            return w, None
        key = self._wkey(w)
        ent = self._packs.get(name)
        if ent is None or ent[0] != key:
            out_f, in_f = w.shape
            bufs = ent[1] if ent is not None else (self._empty(out_f, in_f), self._empty(in_f, out_f))
            check(lib.htrvt_cast_transpose_f32(ptr(w), ptr(bufs[0]), ptr(bufs[1]), out_f, in_f, out_f, self.dti, stream()),
                  "cast_transpose_f32")
            self._packs[name] = (key, bufs)
            return bufs
        return ent[1]

    def _head_w(self, w):
        """head weight in compute dtype, rows zero-padded to a multiple of 8 classes: ([Cp][D], transposed [D][Cp] or None)"""
        C, D = w.shape
        Cp = (C + 7) // 8 * 8
        key = self._wkey(w)
        ent = self._packs.get("head")
        if ent is None or ent[0] != key:
            if self.dtype == torch.float32:
                buf = ent[1][0] if ent is not None else torch.zeros(Cp, D, dtype=self.dtype, device=self.dev)
                buf[:C].copy_(w)          # device memcpy
                bufs = (buf, None)
            else:
                bufs = ent[1] if ent is not None else (torch.zeros(Cp, D, dtype=self.dtype, device=self.dev),
                                                       torch.zeros(D, Cp, dtype=self.dtype, device=self.dev))
                check(lib.htrvt_cast_transpose_f32(ptr(w), ptr(bufs[0]), ptr(bufs[1]), C, D, Cp, self.dti, stream()),
                      "cast_transpose_f32")
            self._packs["head"] = (key, bufs)
            return bufs
        return ent[1]

    def _conv_w(self, name, w):
        """(fwd pack [Co][taps][Cpad_i], dgrad pack [Ci][taps][Cpad_o]) of a conv weight [Co,Ci,k,k]."""
        if self.split:
            return self._conv_w_split(name, w)
        key = self._wkey(w)
        ent = self._packs.get(name)
        Co, Ci, kh, kw = w.shape
        taps = kh * kw
        cpi, cpo = cpad(Ci, self.dtype), cpad(Co, self.dtype)
        if ent is None or ent[0] != key:
            if ent is None:
                fwd = torch.zeros(Co, taps, cpi, dtype=self.dtype, device=self.dev)
                dgr = torch.zeros(Ci, taps, cpo, dtype=self.dtype, device=self.dev)
            else:
                fwd, dgr = ent[1]
            check(lib.htrvt_pack_conv_weight(ptr(w), ptr(fwd), ptr(dgr), Co, Ci, taps, cpi, cpo, self.dti, stream()),
                  "pack_conv_weight")
            self._packs[name] = (key, (fwd, dgr))
            return fwd, dgr
        return ent[1]

    def _conv_w_joint_dgrad(self, name, w3, namd, wd):
        """[Ci][taps + 1][Cpad_o] dgrad pack shared by a block's strided 3x3 conv (tap slots 0 .. taps-1) and its 1x1 downsample
        conv (slot `taps`): the B operand of the parity-class dgrad launches when the downsample gradient rides along as one
        more tap (HtrvtGemmDesc.A2, conv_dgrad(extra=...))."""
        key = (self._wkey(w3), self._wkey(wd))
        ent = self._packs.get(name + "+ds")
        Co, Ci, kh, kw = w3.shape
        taps = kh * kw
        cpi, cpo = cpad(Ci, self.dtype), cpad(Co, self.dtype)
        if ent is None or ent[0] != key:
            buf = ent[1] if ent is not None else torch.zeros(Ci, taps + 1, cpo, dtype=self.dtype, device=self.dev)
            check(lib.htrvt_pack_conv_weight_slots(ptr(w3), None, ptr(buf), Co, Ci, taps, cpi, cpo, taps + 1, 0, self.dti, stream()),
                  "pack_conv_weight_slots")
            check(lib.htrvt_pack_conv_weight_slots(ptr(wd), None, ptr(buf), Co, Ci, 1, cpi, cpo, taps + 1, taps, self.dti, stream()),
                  "pack_conv_weight_slots")
            self._packs[name + "+ds"] = (key, buf)
            return buf
        return ent[1]

    # ------------------------------------------------------------------ GEMM-shaped pieces
    def linear_fwd(self, x, w, bias, out=None, act=0, preact=None, residual=None, c_f32=False):
        M, K = x.shape
        N = w.shape[0]
        if self.split:      # w = (hi | hi | lo) [N][3 K]; float32 in, float32 out
            out = self._empty(M, N) if out is None else out
            xs, _, _ = self._split_act(x, K)
            # the product writes plain float32 + bias (8-phase kernel, 16-byte stores); GELU / saved pre-activation / residual
            # are float32 element-wise passes in the float32 path's order (its GEMM epilogue): v = acc + bias; pre = v;
            # v = gelu(v); v += residual
            first = preact if (act == 1 and preact is not None) else out
            gemm(xs, w, first, dtype=self.gdt, M=M, N=N, K=3 * K, lda=3 * K, ldb=3 * K, ldc=N, bias=bias, c_f32=True)
            if act == 1:
                check(lib.htrvt_elementwise_f32(ptr(first), None, ptr(out), M * N, 0, stream()), "elementwise_f32")
            else:
                assert act == 0 and preact is None
            if residual is not None:
                check(lib.htrvt_elementwise_f32(ptr(out), ptr(residual), ptr(out), M * N, 2, stream()), "elementwise_f32")
            return out
        if out is None:
            out = self._empty(M, N, dtype=torch.float32 if c_f32 else self.dtype)
        gemm(x, w, out, dtype=self.dtype, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, act=act, preact=preact,
             residual=residual, c_f32=c_f32)
        return out

    def linear_dgrad(self, dy, w, wt=None, act=0, preact=None, plain=False):
        """dx[M,K] = dy[M,N] @ w[N,K]  (optionally * gelu'(preact)).  wt = w^T [K][>=N] (bf16 path): K-major B operand."""
        M, N = dy.shape
        if self.split and not plain:      # wt = (hi | hi | lo) of w^T: [K][3 N]
            K = wt.shape[0]
            dx = self._empty(M, K)
            dys, _, _ = self._split_act(dy, N, cat=True)
            gemm(dys, wt, dx, dtype=self.gdt, M=M, N=K, K=3 * N, lda=3 * N, ldb=3 * N, ldc=K, c_f32=True)
            if act == 2:        # * gelu'(saved pre-activation), float32 element-wise
                check(lib.htrvt_elementwise_f32(ptr(dx), ptr(preact), ptr(dx), M * K, 1, stream()), "elementwise_f32")
            else:
                assert act == 0
            return dx
        K = w.shape[1]
        dx = self._empty(M, K)
        if wt is not None:
            gemm(dy, wt, dx, dtype=self.dtype, M=M, N=K, K=N, lda=N, ldb=wt.shape[1], ldc=K, act=act, preact=preact)
        else:
            gemm(dy, w, dx, dtype=self.dtype, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, b_layout=MNMAJOR, act=act, preact=preact)
        return dx

    def _hwgrad_tiles(self, g):
        """(workgroups per pixel range, tile rows, tile columns) of the halo-staged conv weight-gradient kernel, asked of the
        library itself (htrvt_gemm_wgrad_tiling: the same eligibility test htrvt_gemm applies), or None where the generic
        kernel serves the convolution"""
        if self.gdt != torch.bfloat16 or not self.halo_wgrad:
            return None
        import ctypes
        from ._lib import GemmDesc
        d = GemmDesc()
        cp = cpad(g.Ci, self.gdt)
        d.dtype, d.a_layout, d.b_layout, d.gather = dt(self.gdt), MNMAJOR, MNMAJOR, GATHER_CONV_WGRAD
        d.M, d.N, d.K = g.taps * cp, g.Co, g.B * g.Ho * g.Wo
        d.lda, d.ldb, d.ldc = g.Ci, g.Co, g.Co
        g.fill(d)
        d.Cpad, d.c_f32, d.accumulate, d.batch = cp, 1, 1, 1
        tr, tc = ctypes.c_int32(0), ctypes.c_int32(0)
        n = lib.htrvt_gemm_wgrad_tiling(ctypes.byref(d), ctypes.byref(tr), ctypes.byref(tc))
        return (n, tr.value, tc.value) if n > 0 else None

    def _split_k(self, Mo, No, Kred, conv=False, tiling=None):
        """split-K factor of a weight-gradient GEMM: fill the 256 CUs in whole rounds, but keep the float32
        atomic traffic (one full output tile per block, ~1.3 TB/s chip-wide) small against the MFMA time."""
        bm, bn = (256, 192) if self.gdt == torch.bfloat16 else (128, 128)
        if conv and self.gdt == torch.bfloat16 and No % 256 == 0:
            bn = 256      # gemm_dma.hip pick_bn(): conv weight gradients with N % 256 == 0 use 256x256 tiles
        tiles = ((Mo + bm - 1) // bm) * ((No + bn - 1) // bn)
        if tiling is not None:
            tiles, bm, bn = tiling
        flops = 2.0 * Mo * No * Kred
        best, best_t = 1, None
        # multiples of 8 let the kernel keep all tiles of one K range on one XCD (shared L2); small factors otherwise
        # a single output tile (the 1x1 downsample weight gradients, K = 1 M pixels at layer 1): up to one K range per CU
        smax = 257 if tiles == 1 else 129
        # round 5: the split-K kernels keep the (range, tile) pairs of one XCD consecutive for ANY split factor (xcd_range_map,
        # csrc/gemm_dma_impl.h), so every factor up to 64 is a candidate for the conv / MN-major launches, not only the multiples
        # of 8 and a hand-picked few; HTRVT_SPLITK_LEGACY=1 restores the former candidate list and its 10 % penalty (A/B runs)
        legacy = os.environ.get("HTRVT_SPLITK_LEGACY", "0") == "1" or os.environ.get("HTRVT_NO_XCD_RANGES", "0") == "1"
        cands = [1, 2, 3, 4, 5, 6, 7] + list(range(8, smax, 8)) + ([10, 12, 14, 20, 28] if tiling is not None else [])
        if not legacy and (conv or tiling is not None):
            cands = sorted(set(cands) | set(range(8, 65)))
        for s in cands:
            if Kred // s < 512:
                continue
            blocks = tiles * s
            rounds = (blocks + 255) // 256
            # every block's float32 output tile: atomics ~1.3 TB/s chip-wide; slabs are written and read back once at HBM speed
            t = flops / 1.0e15 * (rounds * 256.0 / blocks) + blocks * bm * bn * 4 / (2.5e12 if self.deterministic and s > 1 else 1.3e12)
            if s > 1 and s % 8:
                t *= 1.10 if legacy else 1.02     # (legacy: no XCD grouping of the K ranges, 4-7x the operand traffic; now: a range may straddle two XCDs)
            if best_t is None or t < best_t:
                best, best_t = s, t
        return best

    def _splitk_ws(self, sk, Mo, No):
        if sk > 1 and self.deterministic and No % 4:
            if not getattr(self, "_warned_atomics", False):     # slabs need 16-byte rows: this launch falls back to float atomics
                import warnings
                warnings.warn(f"htrvt_amd: split-K weight gradient with {No} output columns (not a multiple of 4) uses float "
                              "atomics: results are not bitwise reproducible although Engine.deterministic is set")
                self._warned_atomics = True
            return None
        if sk <= 1 or not self.deterministic:
            return None
        return self._empty(sk, Mo, No, dtype=torch.float32)

    # Weight gradients have no consumer inside backward: they run on a side stream, overlapping the
    # (HBM-bound) BatchNorm / LayerNorm backward passes and the tails of the dgrad GEMMs on the main stream.
    def _on_side(self, fn, *tensors):
        if not self._side_active:
            return fn()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        self._side.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            t.record_stream(self._side)      # the caching allocator must not recycle them under the side stream
        with torch.cuda.stream(self._side):
            return fn()

    def _wgrad_stream(self):
        """the stream weight gradients are running on, None when they share the main stream"""
        return self._side if (self._side is not None and self._side_active) else None

    def _join_side(self):
        if self._side is not None and self._side_active:
            torch.cuda.current_stream().wait_stream(self._side)

    def linear_wgrad(self, dy, x, dw, dbias, plain=False):
        return self._on_side(lambda: self._linear_wgrad(dy, x, dw, dbias, plain), dy, x)

    def conv_wgrad(self, dy, x, g, dw):
        return self._on_side(lambda: self._conv_wgrad(dy, x, g, dw), dy, x)

    def _linear_wgrad(self, dy, x, dw, dbias, plain=False):
        """dw[N,K] += dy^T x ; dbias[N] += colsum(dy)."""
        M, N = dy.shape
        K = x.shape[1]
        tiling = None
        if self.gdt == torch.bfloat16 and self.deterministic and not plain:
            # the MN-major 8-phase kernel's 256 x 256 tiles where the LIBRARY says it serves the launch (htrvt_gemm_wgrad_tiling:
            # gemm8pt_serves -- alignment, 2 GiB, the HTRVT_NO_MNMAJOR_8PHASE switch; no copy of that test here)
            import ctypes
            from ._lib import GemmDesc
            d = GemmDesc()
            d.dtype, d.a_layout, d.b_layout, d.gather = dt(self.gdt), MNMAJOR, MNMAJOR, 0
            d.M, d.N, d.K, d.lda, d.ldb, d.ldc = N, K, M, (3 * N if self.split else N), K, K
            d.batch, d.split_k, d.c_f32, d.accumulate = 1, 2, 1, 1
            d.A, d.B, d.C = ptr(dy), ptr(x), ptr(dw)
            tr_, tc_ = ctypes.c_int32(0), ctypes.c_int32(0)
            nt = lib.htrvt_gemm_wgrad_tiling(ctypes.byref(d), ctypes.byref(tr_), ctypes.byref(tc_))
            if nt > 0:
                tiling = (nt, tr_.value, tc_.value)
        sk = self._split_k(N, K, M, tiling=tiling)
        if self.split and not plain:      # the contraction runs over the rows: hi / lo planes, three accumulating launches
            # MN-major plain operands carry their own leading dimension: the hi / lo planes of the gradient are column blocks
            # 0 and 1 of the (hi | lo | hi) form the dgrad launch already made -- no second split of dy
            dy3, _, _ = self._split_act(dy, N, cat=True, planes=False)
            _, xh, xl = self._split_act(x, K, cat=False, planes=True)
            for a_off, b_ in ((0, xh), (0, xl), (N, xh)):
                gemm(dy3, b_, dw, dtype=self.gdt, M=N, N=K, K=M, lda=3 * N, ldb=K, ldc=K, a_layout=MNMAJOR, b_layout=MNMAJOR,
                     split_k=sk, accumulate=True, c_f32=True, splitk_ws=self._splitk_ws(sk, N, K), a_off=a_off)
            if dbias is not None:
                ops.colsum(dy, M, N, N, dbias, dti=self.dti)
            return
        gemm(dy, x, dw, dtype=self.dtype, M=N, N=K, K=M, lda=N, ldb=K, ldc=K, a_layout=MNMAJOR, b_layout=MNMAJOR,
             split_k=sk, accumulate=True, c_f32=True, splitk_ws=self._splitk_ws(sk, N, K))
        if dbias is not None:
            ops.colsum(dy, M, N, N, dbias, dti=self.dti)

    def conv_fwd(self, x, wf, g: ConvGeom, want_stats, bn=None, relu=False, residual=None):
        """bn = (scale, shift): eval-mode BatchNorm (running statistics) folded into the launch -- C = relu?(conv * scale +
        shift [+ residual]) -- instead of a raw conv output plus a BatchNorm pass (train mode needs the batch statistics
        of the whole output first)."""
        M = g.B * g.Ho * g.Wo
        cpi = cpad(g.Ci, self.dtype)
        y = self._empty(g.B, g.Ho, g.Wo, g.Co)
        cs, rows = None, 0
        if want_stats:
            rows = ops.gemm_num_mtiles(M, g.Co, self.gdt, gather=GATHER_CONV_FWD)
            cs = self._empty(rows + 64, 2, g.Co, dtype=torch.float32)
        kw = {}
        if bn is not None:
            kw = dict(colscale=bn[0], bias=bn[1], residual=residual, act=3 if relu else 0)
        if self.split:      # the same convolution over 3 Ci input channels: (hi | lo | hi) pixels against the (hi | hi | lo) pack
            x3, _, _ = self._split_act(x, g.Ci)
            cp3 = cpad(3 * g.Ci, self.gdt)
            # the tripled operand of a layer-1 convolution passes 2 GiB at 128 images (the LDS-DMA kernels address operands
            # through 2 GiB buffer descriptors): one launch per batch slice, each with its own rows of the column sums
            n = self._batch_slices(g.B, x3.numel() * 2, g.Ho * g.Wo)
            Bc = g.B // n
            Mc = Bc * g.Ho * g.Wo
            g3 = ConvGeom(Bc, g.Hi, g.Wi, 3 * g.Ci, g.Co, g.kh, (g.sh, g.sw), g.ph)
            rows_c = ops.gemm_num_mtiles(Mc, g.Co, self.gdt, gather=GATHER_CONV_FWD) if want_stats else 0
            if want_stats:
                rows = n * rows_c
                cs = self._empty(rows + 64, 2, g.Co, dtype=torch.float32)
            for c in range(n):
                kc = dict(kw)
                if kc.get("residual") is not None:
                    kc["residual"] = kc["residual"].view(-1)[c * Mc * g.Co:]
                gemm(x3, wf, y, dtype=self.gdt, M=Mc, N=g.Co, K=g.taps * cp3, lda=3 * g.Ci, ldb=g.taps * cp3, ldc=g.Co,
                     gather=GATHER_CONV_FWD, geom=g3, Cpad=cp3, colstats=cs[c * rows_c:] if want_stats else None, c_f32=True,
                     a_off=c * Bc * g.Hi * g.Wi * 3 * g.Ci, c_off=c * Mc * g.Co, **kc)
            return y, cs, rows
        gemm(x, wf, y, dtype=self.dtype, M=M, N=g.Co, K=g.taps * cpi, lda=g.Ci, ldb=g.taps * cpi, ldc=g.Co,
             gather=GATHER_CONV_FWD, geom=g, Cpad=cpi, colstats=cs, **kw)
        return y, cs, rows

    def _dgrad_merged(self, g, dy=None, wd=None, dx=None, extra=None):
        """M tiles of the merged strided-dgrad launch (HtrvtGemmDesc.cls_h = -2) when the library serves `g` in that form
        (asked of the library itself: htrvt_gemm_dgrad_merged_tiles), else 0"""
        if not (self.merged_strided_dgrad and self.gdt == torch.bfloat16 and not self.split and self._dgrad_by_class(g) and g.kh == 3):
            return 0
        import ctypes
        from ._lib import GemmDesc
        d = GemmDesc()
        cpo = cpad(g.Co, self.dtype)
        d.dtype, d.a_layout, d.b_layout, d.gather = dt(self.dtype), KMAJOR, KMAJOR, GATHER_CONV_DGRAD
        d.M, d.N, d.K = g.B * g.Hi * g.Wi, g.Ci, (g.taps + (1 if extra is not None else 0)) * cpo
        d.lda, d.ldb, d.ldc = g.Co, (g.taps + (1 if extra is not None else 0)) * cpo, g.Ci
        g.fill(d)
        d.Cpad, d.batch, d.split_k, d.cls_h, d.cls_w = cpo, 1, 1, -2, -2
        # the pointer tests of the eligibility check (alignment, A2 behind A): the real ones when known, aligned stand-ins else
        d.A = ptr(dy) if dy is not None else 4096
        d.B = ptr(wd) if wd is not None else 4096
        d.C = ptr(dx) if dx is not None else 4096
        d.A2 = ptr(extra)
        return int(lib.htrvt_gemm_dgrad_merged_tiles(ctypes.byref(d)))

    def dgrad_tiles(self, g: ConvGeom):
        """number of M tiles (rows of a fused BN-backward partial buffer) conv_dgrad will produce"""
        nm = self._dgrad_merged(g)
        if nm > 0:
            return nm
        if self._dgrad_by_class(g):
            return sum(ops.gemm_num_mtiles(g.B * ((g.Hi - a + g.sh - 1) // g.sh) * ((g.Wi - b + g.sw - 1) // g.sw), g.Ci,
                                           self.dtype, gather=GATHER_CONV_DGRAD) for a in range(g.sh) for b in range(g.sw))
        return ops.gemm_num_mtiles(g.B * g.Hi * g.Wi, g.Ci, self.dtype, gather=GATHER_CONV_DGRAD)

    def _dgrad_by_class(self, g):
        return self.gdt == torch.bfloat16 and (g.sh, g.sw) != (1, 1) and g.B * g.Hi * g.Wi // (g.sh * g.sw) > 128

    def conv_dgrad(self, dy, wd, g: ConvGeom, residual=None, relu_src=None, bnb=None, extra=None, relu_bn=None, relu_bits=False):
        """dx = conv-dgrad(dy) [+ residual] [masked by relu_src > 0]; bnb: fused BatchNorm-backward sums (bf16 only).
        extra = dy2 (parity-class path only): the gradient of the block's 1x1 downsample conv output, allocated right
        behind dy; wd is then the joint pack of _conv_w_joint_dgrad and dx also receives the 1x1 conv's input gradient."""
        cpo = cpad(g.Co, self.dtype)
        dx = self._empty(g.B, g.Hi, g.Wi, g.Ci)
        assert extra is None or self._dgrad_by_class(g)
        assert relu_bn is None or (not self._dgrad_by_class(g) and relu_src is None and residual is None and bnb is not None and len(bnb) == 1)
        wtaps = g.taps + (1 if extra is not None else 0)       # taps per row of the packed weight
        if self._dgrad_by_class(g) and self.split:
            # split-bf16: the same parity-class launches over 3 Co gradient channels; each leaves a dense float32 [B,Hq,Wq,Ci]
            # matrix (the float32 epilogue of the LDS-DMA kernels stores class rows as they come), one pass interleaves the
            # classes into dx and adds the residual (htrvt_class_scatter_f32)
            assert relu_src is None and bnb is None and relu_bn is None and extra is None
            dy3, _, _ = self._split_act(dy, g.Co, cat=True, planes=True)
            cp3 = cpad(3 * g.Co, self.gdt)
            g3 = ConvGeom(g.B, g.Hi, g.Wi, g.Ci, 3 * g.Co, g.kh, (g.sh, g.sw), g.ph)
            parts = {}
            for a in range(g.sh):
                for b in range(g.sw):
                    nt = sum(1 for dy_ in range(g.kh) if (a + g.ph - dy_) % g.sh == 0) * \
                         sum(1 for dx_ in range(g.kw) if (b + g.pw - dx_) % g.sw == 0)
                    Hq, Wq = (g.Hi - a + g.sh - 1) // g.sh, (g.Wi - b + g.sw - 1) // g.sw
                    if nt == 0:      # no tap reaches this class (1x1 strided conv): its gradient is zero
                        parts[(a, b)] = torch.zeros(g.B, Hq, Wq, g.Ci, dtype=torch.float32, device=self.dev)
                        continue
                    part = self._empty(g.B, Hq, Wq, g.Ci)
                    # cls selects the taps and the gathered pixels; with cls the float32 epilogue writes row m of the class
                    # at row m of C: a dense matrix
                    gemm(dy3, wd, part, dtype=self.gdt, M=g.B * Hq * Wq, N=g.Ci, K=nt * cp3, lda=3 * g.Co, ldb=g.taps * cp3, ldc=g.Ci,
                         gather=GATHER_CONV_DGRAD, geom=g3, Cpad=cp3, cls=(a, b), c_f32=True)
                    parts[(a, b)] = part
            check(lib.htrvt_class_scatter_f32(ptr(parts[(0, 0)]), ptr(parts.get((0, 1))), ptr(parts.get((1, 0))), ptr(parts.get((1, 1))),
                                              ptr(residual), ptr(dx), g.B, g.Hi, g.Wi, g.Ci, g.sh, g.sw, stream()), "class_scatter_f32")
            return dx
        if self._dgrad_merged(g, dy, wd, dx, extra) > 0:
            # strided 3x3 conv: every parity class in one launch, halo-staged tiles; the downsample gradient rides along (A2)
            gemm(dy, wd, dx, dtype=self.dtype, M=g.B * g.Hi * g.Wi, N=g.Ci, K=wtaps * cpo, lda=g.Co, ldb=wtaps * cpo, ldc=g.Ci,
                 gather=GATHER_CONV_DGRAD, geom=g, Cpad=cpo, residual=residual, cls=(-2, -2), relu_src=relu_src,
                 relu_bits=relu_bits, bnb=bnb, a2=extra)
            return dx
        if self._dgrad_by_class(g):
            # strided conv: one launch per input-pixel parity class, each contracting only the taps that reach it
            tile0 = 0
            par = self.parallel_classes and ops.PROFILE is None
            main = torch.cuda.current_stream()
            if par and self._cls_streams is None:
                self._cls_streams = [torch.cuda.Stream(device=self.dev) for _ in range(3)]
            used = []
            for a in range(g.sh):
                for b in range(g.sw):
                    nt = sum(1 for dy_ in range(g.kh) if (a + g.ph - dy_) % g.sh == 0) * \
                         sum(1 for dx_ in range(g.kw) if (b + g.pw - dx_) % g.sw == 0)
                    Hq, Wq = (g.Hi - a + g.sh - 1) // g.sh, (g.Wi - b + g.sw - 1) // g.sw
                    idx = a * g.sw + b
                    st_ = self._cls_streams[idx - 1] if (par and idx > 0) else main
                    if st_ is not main:
                        st_.wait_stream(main)
                        used.append(st_)
                    a2 = extra if (extra is not None and (a, b) == (0, 0)) else None    # the pixels a 1x1 stride-s conv reads
                    with torch.cuda.stream(st_):
                        gemm(dy, wd, dx, dtype=self.dtype, M=g.B * Hq * Wq, N=g.Ci, K=(nt + (1 if a2 is not None else 0)) * cpo,
                             lda=g.Co, ldb=wtaps * cpo, ldc=g.Ci, gather=GATHER_CONV_DGRAD, geom=g, Cpad=cpo, residual=residual,
                             cls=(a, b), relu_src=relu_src, relu_bits=relu_bits, bnb=bnb, bnb_tile0=tile0, a2=a2)
                    tile0 += ops.gemm_num_mtiles(g.B * Hq * Wq, g.Ci, self.dtype, gather=GATHER_CONV_DGRAD)
            for st_ in used:       # every class has written its pixels before anything downstream reads dx
                main.wait_stream(st_)
            return dx
        if self.split:      # 3 Co gradient channels: (hi | lo | hi) against the (hi | hi | lo) dgrad pack
            assert relu_src is None and bnb is None and relu_bn is None
            dy3, _, _ = self._split_act(dy, g.Co, cat=True, planes=True)
            cp3 = cpad(3 * g.Co, self.gdt)
            n = self._batch_slices(g.B, dy3.numel() * 2, g.Hi * g.Wi)
            Bc = g.B // n
            Mc = Bc * g.Hi * g.Wi
            g3 = ConvGeom(Bc, g.Hi, g.Wi, g.Ci, 3 * g.Co, g.kh, (g.sh, g.sw), g.ph)
            for c in range(n):
                gemm(dy3, wd, dx, dtype=self.gdt, M=Mc, N=g.Ci, K=g.taps * cp3, lda=3 * g.Co, ldb=g.taps * cp3,
                     ldc=g.Ci, gather=GATHER_CONV_DGRAD, geom=g3, Cpad=cp3,
                     residual=None if residual is None else residual.view(-1)[c * Mc * g.Ci:], c_f32=True,
                     a_off=c * Bc * g.Ho * g.Wo * 3 * g.Co, c_off=c * Mc * g.Ci)
            return dx
        gemm(dy, wd, dx, dtype=self.dtype, M=g.B * g.Hi * g.Wi, N=g.Ci, K=g.taps * cpo, lda=g.Co, ldb=g.taps * cpo,
             ldc=g.Ci, gather=GATHER_CONV_DGRAD, geom=g, Cpad=cpo, residual=residual, relu_src=relu_src, relu_bits=relu_bits, bnb=bnb,
             relu_bn=relu_bn)
        return dx

    def _conv_wgrad(self, dy, x, g: ConvGeom, dw):
        M = g.B * g.Ho * g.Wo
        cpi = cpad(g.Ci, self.gdt)
        packed = self._zeros(g.taps, cpi, g.Co)
        sk = self._split_k(g.taps * cpi, g.Co, M, conv=True, tiling=self._hwgrad_tiles(g))
        if self.split:      # the contraction runs over the pixels: hi / lo planes, three accumulating launches
            _, xh, xl = self._split_act(x, g.Ci, cat=False, planes=True)
            _, dyh, dyl = self._split_act(dy, g.Co, cat=False, planes=True)
            pairs = ((xh, dyh), (xl, dyh), (xh, dyl))
        else:
            pairs = ((x, dy),)
        for x_, dy_ in pairs:
            gemm(x_, dy_, packed, dtype=self.gdt, M=g.taps * cpi, N=g.Co, K=M, lda=g.Ci, ldb=g.Co, ldc=g.Co,
                 a_layout=MNMAJOR, b_layout=MNMAJOR, gather=GATHER_CONV_WGRAD, geom=g, Cpad=cpi,
                 split_k=sk, accumulate=True, c_f32=True, splitk_ws=self._splitk_ws(sk, g.taps * cpi, g.Co))
        if self.table_relayout and self._zoff is not None:    # inside backward(): unpacked with its DP bucket, one launch
            self._pending_unpack.append((packed, dw, g.Co, g.Ci, g.taps, cpi))
        else:
            check(lib.htrvt_unpack_conv_wgrad(ptr(packed), ptr(dw), g.Co, g.Ci, g.taps, cpi, stream()), "unpack_conv_wgrad")

    def _flush_unpacks(self):
        """grad [Co][Ci][taps] += every pending conv weight-gradient GEMM output, one launch, on the stream that produced them"""
        if not self._pending_unpack:
            return
        pend, self._pending_unpack = self._pending_unpack, []

        def run():
            self._relayout([_job(RELAYOUT_UNPACK_WGRAD, packed, dw, None, Co, Ci, taps, cpi) for packed, dw, Co, Ci, taps, cpi in pend])
        self._on_side(run)

    # ------------------------------------------------------------------ table-driven re-layouts (csrc/relayout.hip)
    def _relayout(self, jobs):
        """run RelayoutJobs, <= 48 per launch: the table is planned on the host and travels in the kernel-argument segment
        (no device copy to keep alive, nothing to refresh when autograd hands out new gradient buffers)"""
        for i in range(0, len(jobs), RELAYOUT_ARG_JOBS):
            chunk = jobs[i:i + RELAYOUT_ARG_JOBS]
            arr = (RelayoutJob * len(chunk))(*chunk)
            total = lib.htrvt_relayout_plan(arr, len(chunk))
            if total < 0:
                raise RuntimeError(lib.htrvt_last_error().decode())
            check(lib.htrvt_relayout_host(arr, len(chunk), total, self.dti, stream()), "relayout")

    def _repack_all(self, P, save):
        """bf16: refresh every stale conv pack / Linear copy in one launch (what _conv_w / _conv_w_joint_dgrad / _lin_w /
        _head_w would do one tensor at a time on their first use after an optimizer step)."""
        s = self.s
        jobs, fresh = [], []
        for name, _ci, _co, _k, _st, _pd in s.stem_convs():
            w = P[name + ".weight"]
            key, ent = self._wkey(w), self._packs.get(name)
            if ent is not None and ent[0] == key:
                continue
            Co, Ci, kh, kw = w.shape
            taps, cpi, cpo = kh * kw, cpad(Ci, self.dtype), cpad(Co, self.dtype)
            bufs = ent[1] if ent is not None else (torch.zeros(Co, taps, cpi, dtype=self.dtype, device=self.dev),
                                                   torch.zeros(Ci, taps, cpo, dtype=self.dtype, device=self.dev))
            jobs.append(_job(RELAYOUT_PACK_CONV, w, bufs[0], bufs[1], Co, Ci, taps, cpi, cpo, taps, 0))
            fresh.append((name, key, bufs))
        if save and self.fuse_downsample_dgrad:
            for li in (1, 2, 3):
                pb = f"patch_embed.layer{li}.0"
                w3, wd = P[pb + ".conv1.weight"], P[pb + ".downsample.0.weight"]
                key, ent = (self._wkey(w3), self._wkey(wd)), self._packs.get(pb + ".conv1+ds")
                if ent is not None and ent[0] == key:
                    continue
                Co, Ci, kh, kw = w3.shape
                taps, cpi, cpo = kh * kw, cpad(Ci, self.dtype), cpad(Co, self.dtype)
                buf = ent[1] if ent is not None else torch.zeros(Ci, taps + 1, cpo, dtype=self.dtype, device=self.dev)
                jobs.append(_job(RELAYOUT_PACK_CONV, w3, None, buf, Co, Ci, taps, cpi, cpo, taps + 1, 0))
                jobs.append(_job(RELAYOUT_PACK_CONV, wd, None, buf, Co, Ci, 1, cpi, cpo, taps + 1, taps))
                fresh.append((pb + ".conv1+ds", key, buf))
        for name in s.linears():
            w = P[name + ".weight"]
            key, ent = self._wkey(w), self._packs.get(name)
            if ent is not None and ent[0] == key:
                continue
            out_f, in_f = w.shape
            ld_t = (out_f + 7) // 8 * 8 if name == "head" else out_f     # head: classes zero-padded to a multiple of 8
            if ent is not None:
                bufs = ent[1]
            elif name == "head":
                bufs = (torch.zeros(ld_t, in_f, dtype=self.dtype, device=self.dev), torch.zeros(in_f, ld_t, dtype=self.dtype, device=self.dev))
            else:
                bufs = (self._empty(out_f, in_f), self._empty(in_f, out_f))
            jobs.append(_job(RELAYOUT_CAST_TRANSPOSE, w, bufs[0], bufs[1], out_f, in_f, 0, ld_t))
            fresh.append((name, key, bufs))
        if jobs:
            self._relayout(jobs)
            for name, key, bufs in fresh:
                self._packs[name] = (key, bufs)

    # ------------------------------------------------------------------ BatchNorm pieces
    def bn_coeffs(self, P, prefix, C, train, cs=None, rows=0, count=0, save=False):
        """returns (scale, shift, save_mean, save_rstd)"""
        scale, shift = self._empty(C, dtype=torch.float32), self._empty(C, dtype=torch.float32)
        if train:
            mean, rstd = self._empty(C, dtype=torch.float32), self._empty(C, dtype=torch.float32)
            check(lib.htrvt_bn_finalize(ptr(cs), rows, C, float(count), ptr(P[prefix + ".weight"]), ptr(P[prefix + ".bias"]),
                                        BN_EPS, BN_MOMENTUM, ptr(P[prefix + ".running_mean"]),
                                        ptr(P[prefix + ".running_var"]), ptr(P[prefix + ".num_batches_tracked"]),
                                        ptr(scale), ptr(shift), ptr(mean), ptr(rstd), stream()), "bn_finalize")
            return scale, shift, mean, rstd
        # eval mode (running statistics).  save: an eval-mode backward (frozen-BN fine-tuning, saliency) needs them as
        # (mean, rstd); they are constants there, bn_backward_finish is told so through self._bn_train
        rstd = self._empty(C, dtype=torch.float32) if save else None
        check(lib.htrvt_bn_eval_coeffs(ptr(P[prefix + ".weight"]), ptr(P[prefix + ".bias"]), ptr(P[prefix + ".running_mean"]),
                                       ptr(P[prefix + ".running_var"]), BN_EPS, ptr(scale), ptr(shift), ptr(rstd), C, stream()),
              "bn_eval_coeffs")
        return scale, shift, (P[prefix + ".running_mean"] if save else None), rstd

    def bn_apply(self, x, scale, shift, relu, res=None, rscale=None, rshift=None, want_mask=False):
        """y = [relu](x * scale + shift [+ res [* rscale + rshift]]); want_mask: also the 1-bit-per-element sign mask of y
        (uint8 [numel / 8]) that the backward of this ReLU reads instead of y (HtrvtGemmDesc.relu_bits)"""
        y = torch.empty_like(x)
        C = x.shape[-1]
        if want_mask:
            mask = torch.empty(x.numel() // 8, dtype=torch.uint8, device=x.device)
            check(lib.htrvt_bn_apply_mask(ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(rscale), ptr(rshift), ptr(y), ptr(mask),
                                          x.numel() // C, C, 1 if relu else 0, self.dti, stream()), "bn_apply_mask")
            return y, mask
        check(lib.htrvt_bn_apply(ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(rscale), ptr(rshift), ptr(y),
                                 x.numel() // C, C, 1 if relu else 0, self.dti, stream()), "bn_apply")
        return y

    def bn_backward(self, dy, yact, x, prefix, P, G, mean, rstd, want_g=False, out=None):
        """dx of train-mode BN (+ReLU mask from yact); accumulates dgamma/dbeta into G.  Unfused form: one
        reduction pass over (dy, yact, x), then bn_backward_finish."""
        C = x.shape[-1]
        npix = x.numel() // C
        nblk = lib.htrvt_bn_bwd_blocks(npix)
        partial = self._empty(nblk, 2, C, dtype=torch.float32)
        check(lib.htrvt_bn_bwd_reduce(ptr(dy), ptr(yact), ptr(x), ptr(mean), ptr(rstd), ptr(partial), npix, C, self.dti,
                                      stream()), "bn_bwd_reduce")
        return self.bn_backward_finish(partial, nblk, dy, yact, x, prefix, P, G, mean, rstd, want_g, out=out)

    def bn_backward_coef(self, partial, rows, x, prefix, P, G, mean, rstd):
        """finalize dgamma / dbeta (accumulated into G) and the [3][C] coefficients of dx = cA*g + cB*x + cC from per-tile
        partial sums"""
        C = x.shape[-1]
        npix = x.numel() // C
        coef = self._empty(3, C, dtype=torch.float32)
        src = partial
        if rows > 64:   # two-level reduction of the partial rows
            red = self._zeros(2 * C)
            ops.colsum(partial, rows, 2 * C, 2 * C, red, dti=0)
            src, rows = red, 1
        count = float(npix) if self._bn_train else -1.0      # eval mode: dx = gamma * rstd * g
        check(lib.htrvt_bn_bwd_finalize(ptr(src), rows, C, count, ptr(P[prefix + ".weight"]), ptr(mean), ptr(rstd),
                                        ptr(G[prefix + ".weight"]), ptr(G[prefix + ".bias"]), ptr(coef), stream()),
              "bn_bwd_finalize")
        return coef

    def bn_backward_finish(self, partial, rows, g, yact, x, prefix, P, G, mean, rstd, want_g=False, out=None):
        """finalize (dgamma, dbeta, coefficients) from per-tile partial sums, then dx = cA*g + cB*x + cC.
        With yact=None, g is the already ReLU-masked gradient (fused dgrad epilogue)."""
        C = x.shape[-1]
        npix = x.numel() // C
        coef = self.bn_backward_coef(partial, rows, x, prefix, P, G, mean, rstd)
        dx = torch.empty_like(x) if out is None else out
        gout = torch.empty_like(x) if want_g else None
        check(lib.htrvt_bn_bwd_apply(ptr(g), ptr(yact), ptr(x), ptr(coef), ptr(dx), ptr(gout), npix, C, self.dti, stream()),
              "bn_bwd_apply")
        return dx, gout

    # ------------------------------------------------------------------ LayerNorm pieces
    def ln_fwd(self, x, gamma, beta, save):
        rows, D = x.shape
        y = torch.empty_like(x)
        mean = self._empty(rows, dtype=torch.float32) if save else None
        rstd = self._empty(rows, dtype=torch.float32) if save else None
        check(lib.htrvt_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, D, self.s.ln_eps, self.dti,
                                      stream()), "layernorm_fwd")
        return y, mean, rstd

    def ln_bwd(self, dy, x, mean, rstd, gamma, dres, dgamma, dbeta):
        rows, D = x.shape
        nblk = lib.htrvt_layernorm_bwd_blocks(rows)
        partial = self._empty(nblk, 2, D, dtype=torch.float32)
        dx = torch.empty_like(x)
        check(lib.htrvt_layernorm_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx), ptr(partial),
                                      rows, D, self.dti, stream()), "layernorm_bwd")
        if dbeta.data_ptr() == dgamma.data_ptr() + 4 * D:   # weight and bias adjacent in the flat gradient buffer: one launch
            ops.colsum(partial, nblk, 2 * D, 2 * D, dgamma, dti=0)
        else:
            ops.colsum(partial, nblk, D, 2 * D, dgamma, dti=0)
            ops.colsum(partial.data_ptr() + 4 * D, nblk, D, 2 * D, dbeta, dti=0)
        return dx

    # ------------------------------------------------------------------ forward
    def forward(self, P, img, keep_mask=None, train=False, save=False):
        """P: dict name -> float32 device tensor (parameters and BN buffers, reference state_dict names).
        img: [B,1,H,W] float32.  keep_mask: None or float32 [N] (1 keep / 0 mask-token).
        Returns float32 logits [B,N,nb_cls] (after the final param-free LayerNorm)."""
        s = self.s
        assert img.is_cuda and img.dtype in (torch.float32, torch.uint8) and img.is_contiguous()
        u8 = 1 if img.dtype == torch.uint8 else 0    # uint8 pixels are read as value / 255 (ToTensor) by the first kernels
        B, _, H, W = img.shape
        assert (H, W) == (s.H, s.W), f"model built for {s.H}x{s.W}, got {H}x{W}"
        # Host run-ahead throttle: enqueueing a step takes ~6 ms of host time against ~40 ms on the device, and nothing in
        # a step makes the host wait.  Unbounded, the host queues many steps; every step's activations are then alive at
        # once (the caching allocator cannot reuse a block before the streams that touched it have passed its
        # record_stream events), the pool grows from 20 to ~96 GiB and the hundreds of device allocations that takes land
        # in the middle of the run (measured: 120 ms per step for the first ten steps after warm-up).  So: before
        # enqueueing call k the host waits until the device has STARTED call k-1 -- one call of run-ahead, which is all
        # the device needs to never run dry.
        # (while a HIP graph is being captured -- Trainer.capture_step -- nothing may wait on the host and there is no
        # run-ahead to bound: the replaying caller throttles itself)
        if not self.capturing:
            if self._call_started is not None:
                self._call_started.synchronize()
            self._call_started = torch.cuda.Event()
            self._call_started.record()
        st = stream()
        sv = {} if save else None
        self._split_cache = []
        self._saved_planes, self._saving = {}, bool(save) and self.split
        C1 = s.D // 4
        keep = None
        if keep_mask is not None:   # uploaded before anything is enqueued: a pageable host->device copy waits for the stream
            keep = keep_mask.to(dtype=torch.float32).contiguous()
            if not keep.is_cuda:    # through pinned memory: a pageable copy would make the host wait for the previous step
                keep = keep.pin_memory().to(self.dev, non_blocking=True)

        prefetched = False
        if self.prefetch_packs and self.dtype == torch.bfloat16 and self.single_stream:
            self._repack_all(P, save) if self.table_relayout else None
        elif self.prefetch_packs and self.dtype == torch.bfloat16:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.dev)
            self._side.wait_stream(torch.cuda.current_stream())     # behind whatever wrote the weights (the optimizer step)
            with torch.cuda.stream(self._side):
                if self.table_relayout:
                    self._repack_all(P, save)
                else:
                    for name, _ci, _co, _k, _st, _pd in s.stem_convs():
                        self._conv_w(name, P[name + ".weight"])
                    if save and self.fuse_downsample_dgrad:
                        for li in (1, 2, 3):
                            pb = f"patch_embed.layer{li}.0"
                            self._conv_w_joint_dgrad(pb + ".conv1", P[pb + ".conv1.weight"], pb + ".downsample.0", P[pb + ".downsample.0.weight"])
                    for name in s.linears():
                        if name == "head":
                            self._head_w(P["head.weight"])
                        else:
                            self._lin_w(name, P[name + ".weight"])
            prefetched = True

        # --- whitening statistics + conv1 + BN + ReLU + maxpool (resnet18.py:74-77) ---
        stats = self._empty(B, 2, dtype=torch.float32)
        check(lib.htrvt_img_stats(ptr(img), ptr(stats), B, H * W, WHITEN_EPS, u8, st), "img_stats")
        w1 = P["patch_embed.conv1.weight"]
        Hp = (H // 2 - 1) // 2 + 1
        a = self._empty(B, Hp, W, C1)
        idx = torch.empty(B, Hp, W, C1, dtype=torch.uint8, device=self.dev) if save else None
        # The conv1 tensor (1.6 GB at the bench shape) is needed only by the unfused backward of the stem; otherwise
        # conv1 + BatchNorm + ReLU + max-pool run as ONE pass over the image, the batch statistics of a train-mode
        # BatchNorm coming from the image's second moments (csrc/stem.hip: stem_moments / stem_stats / stem_fused_fwd).
        fused_stem = self.fuse_stem_forward and (not save or (train and self.fuse_conv1_backward))
        if fused_stem:
            c1 = None
            if train:
                cs = self._empty(2, C1, dtype=torch.float32)
                part = self._empty(lib.htrvt_stem_stats_rows(B, H), 64, dtype=torch.float32)
                check(lib.htrvt_stem_stats(ptr(img), ptr(stats), ptr(w1), ptr(part), ptr(cs), B, H, W, C1, u8, st), "stem_stats")
                sc, sf, mean, rstd = self.bn_coeffs(P, "patch_embed.bn1", C1, True, cs, 1, B * (H // 2) * W, save=save)
            else:
                sc, sf, mean, rstd = self.bn_coeffs(P, "patch_embed.bn1", C1, False, save=save)
            check(lib.htrvt_stem_fwd(ptr(img), ptr(stats), ptr(w1), ptr(sc), ptr(sf), ptr(a), ptr(idx), B, H, W, C1, self.dti,
                                     u8, st), "stem_fwd")
        else:
            c1 = self._empty(B, H // 2, W, C1)
            cs = self._empty(B * (H // 2) + 64, 2, C1, dtype=torch.float32) if train else self._empty(B * (H // 2), 2, C1, dtype=torch.float32)
            check(lib.htrvt_conv1_fwd(ptr(img), ptr(stats), ptr(w1), ptr(c1), ptr(cs), B, H, W, C1, self.dti, u8, st), "conv1_fwd")
            sc, sf, mean, rstd = self.bn_coeffs(P, "patch_embed.bn1", C1, train, cs, B * (H // 2), B * (H // 2) * W, save=save)
            check(lib.htrvt_bn_relu_maxpool(ptr(c1), ptr(sc), ptr(sf), ptr(a), ptr(idx), B, H // 2, W, C1, self.dti, st),
                  "bn_relu_maxpool")
        if save:
            sv["img"], sv["stats"], sv["c1"], sv["bn1"], sv["idx"] = img, stats, c1, (sc, sf, mean, rstd), idx
            sv["c1_shape"] = (B, H // 2, W, C1)

        if prefetched:
            torch.cuda.current_stream().wait_stream(self._side)

        # --- residual stages (resnet18.py:79-81, 23-39) ---
        x = a
        Hc, Wc, Cin = Hp, W, C1
        blocks_saved = []
        for li, (planes, stride) in enumerate(((s.D // 4, (2, 1)), (s.D // 2, (2, 2)), (s.D, (2, 2))), start=1):
            for bi in range(2):
                p = f"patch_embed.layer{li}.{bi}"
                strd = stride if bi == 0 else (1, 1)
                g1 = ConvGeom(B, Hc, Wc, Cin, planes, 3, strd, 1)
                wf1, _ = self._conv_w(p + ".conv1", P[p + ".conv1.weight"])
                g2 = ConvGeom(B, g1.Ho, g1.Wo, planes, planes, 3, (1, 1), 1)
                wf2, _ = self._conv_w(p + ".conv2", P[p + ".conv2.weight"])
                if not train and not save:
                    # eval: running statistics are known up front, every BatchNorm (+ residual + ReLU) rides in the
                    # epilogue of its convolution: 2-3 launches per block, no normalisation passes
                    bn_a = self.bn_coeffs(P, p + ".bn1", planes, False)
                    a1, _, _ = self.conv_fwd(x, wf1, g1, False, bn=bn_a, relu=True)
                    bn_b = self.bn_coeffs(P, p + ".bn2", planes, False)
                    res = x
                    if bi == 0:
                        gd = ConvGeom(B, Hc, Wc, Cin, planes, 1, strd, 0)
                        wfd, _ = self._conv_w(p + ".downsample.0", P[p + ".downsample.0.weight"])
                        bn_d = self.bn_coeffs(P, p + ".downsample.1", planes, False)
                        res, _, _ = self.conv_fwd(x, wfd, gd, False, bn=bn_d)
                    out, _, _ = self.conv_fwd(a1, wf2, g2, False, bn=bn_b, relu=True, residual=res)
                    x = out
                    Hc, Wc, Cin = g1.Ho, g1.Wo, planes
                    continue
                ca, cs1, r1 = self.conv_fwd(x, wf1, g1, train)
                bn_a = self.bn_coeffs(P, p + ".bn1", planes, train, cs1, r1, B * g1.Ho * g1.Wo, save=save)
                a1 = self.bn_apply(ca, bn_a[0], bn_a[1], relu=True)
                cb, cs2, r2 = self.conv_fwd(a1, wf2, g2, train)
                bn_b = self.bn_coeffs(P, p + ".bn2", planes, train, cs2, r2, B * g2.Ho * g2.Wo, save=save)
                if bi == 0:
                    gd = ConvGeom(B, Hc, Wc, Cin, planes, 1, strd, 0)
                    wfd, _ = self._conv_w(p + ".downsample.0", P[p + ".downsample.0.weight"])
                    cd, csd, rd = self.conv_fwd(x, wfd, gd, train)
                    bn_d = self.bn_coeffs(P, p + ".downsample.1", planes, train, csd, rd, B * gd.Ho * gd.Wo, save=save)
                    res_kw = dict(res=cd, rscale=bn_d[0], rshift=bn_d[1])
                else:
                    gd, cd, bn_d = None, None, None
                    res_kw = dict(res=x)
                # the block's output ReLU: its backward (fused into the NEXT block's conv1 dgrad) reads one bit per element
                want_mask = bool(save and self.relu_bitmask and self.fuse_bn_backward and self.dtype == torch.bfloat16 and planes % 8 == 0)
                out = self.bn_apply(cb, bn_b[0], bn_b[1], relu=True, want_mask=want_mask, **res_kw)
                out, omask = out if want_mask else (out, None)
                if save:
                    blocks_saved.append(dict(p=p, x=x, g1=g1, ca=ca, bn_a=bn_a, a1=a1, g2=g2, cb=cb, bn_b=bn_b, gd=gd, cd=cd,
                                             bn_d=bn_d, out=out, mask=omask))
                x = out
                Hc, Wc, Cin = g1.Ho, g1.Wo, planes

        # --- final maxpool + span mask + pos-embed -> tokens (resnet18.py:82, HTR_VT.py:226-231) ---
        Ht = (Hc - 1) // 2 + 1
        N = Ht * Wc
        assert N == s.num_patches, f"token count {N} != num_patches {s.num_patches}"
        D = s.D
        tok = self._empty(B, N, D)
        pos = P["pos_embed"].reshape(N, D)
        check(lib.htrvt_pool_tokens(ptr(x), ptr(keep), ptr(P["mask_token"]), ptr(pos), ptr(tok), B, Hc, N, D, self.dti, st),
              "pool_tokens")
        if save:
            sv["stem_blocks"], sv["l3"], sv["keep"], sv["l3_shape"] = blocks_saved, x, keep, (B, Hc, Wc)

        # --- transformer blocks (HTR_VT.py:80-83, 27-39) ---
        M = B * N
        h, hd = s.heads, s.hd
        scale = hd ** -0.5
        xt = tok.view(M, D)
        enc_saved = []
        for i in range(s.depth):
            p = f"blocks.{i}"
            ln1, m1, r1 = self.ln_fwd(xt, P[p + ".norm1.weight"], P[p + ".norm1.bias"], save)
            wq, _ = self._lin_w(p + ".attn.qkv", P[p + ".attn.qkv.weight"])
            qkv = self.linear_fwd(ln1, wq, P[p + ".attn.qkv.bias"])
            O = self._empty(M, D)
            if self.fused_attention and lib.htrvt_attn_supported(N, hd, self.dti):
                # bf16: one launch, scores / probabilities stay on chip; lse2 is what the recomputing backward needs
                Pm, lse = None, (self._empty(B * h, N, dtype=torch.float32) if save else None)
                check(lib.htrvt_attn_fwd(ptr(qkv), None, ptr(O), ptr(lse), B, N, h, hd, scale, self.dti, st), "attn_fwd")
            else:
                Pm, lse = self._attention_fwd_unfused(qkv, O, B, N, D, h, hd, scale, st), None
            wp, _ = self._lin_w(p + ".attn.proj", P[p + ".attn.proj.weight"])
            x1 = self.linear_fwd(O, wp, P[p + ".attn.proj.bias"], residual=xt)
            ln2, m2, r2 = self.ln_fwd(x1, P[p + ".norm2.weight"], P[p + ".norm2.bias"], save)
            w1_, _ = self._lin_w(p + ".mlp.fc1", P[p + ".mlp.fc1.weight"])
            hpre = self._empty(M, s.hidden) if save else None
            hact = self.linear_fwd(ln2, w1_, P[p + ".mlp.fc1.bias"], act=1, preact=hpre)
            w2_, _ = self._lin_w(p + ".mlp.fc2", P[p + ".mlp.fc2.weight"])
            x2 = self.linear_fwd(hact, w2_, P[p + ".mlp.fc2.bias"], residual=x1)
            if save:
                enc_saved.append(dict(p=p, x0=xt, ln1=ln1, m1=m1, r1=r1, qkv=qkv, P=Pm, lse=lse, O=O, x1=x1, ln2=ln2, m2=m2, r2=r2,
                                      hpre=hpre, h=hact))
            xt = x2

        # --- norm + head + sequence LayerNorm (HTR_VT.py:236-239) ---
        xn, mn, rn = self.ln_fwd(xt, P["norm.weight"], P["norm.bias"], save)
        wh, _ = self._head_w(P["head.weight"])
        raw = self._empty(M, s.nb_cls, dtype=torch.float32)
        gemm(xn, wh, raw, dtype=self.dtype, M=M, N=s.nb_cls, K=D, lda=D, ldb=D, ldc=s.nb_cls, bias=P["head.bias"], c_f32=True)
        y = self._empty(B, N, s.nb_cls, dtype=torch.float32)
        sstats = self._empty(B, 2, dtype=torch.float32)
        check(lib.htrvt_seq_whiten_fwd(ptr(raw), ptr(y), ptr(sstats), B, N * s.nb_cls, WHITEN_EPS, 0, st), "seq_whiten_fwd")
        self._saving = False
        if save:
            sv.update(enc=enc_saved, x_last=xt, xn=xn, mn=mn, rn=rn, y=y, sstats=sstats, B=B, N=N, train=train)
            self.saved = sv
        return y

    def _attention_fwd_unfused(self, qkv, O, B, N, D, h, hd, scale, st):
        """float32 path (and shapes the fused kernel does not serve): S = scale q k^T, row softmax, O = P v as batched
        GEMMs over the [B,N,3,h,hd] layout; returns P (kept for the backward)"""
        S = self._empty(B * h, N, N, dtype=torch.float32)
        gemm(qkv, qkv, S, dtype=self.dtype, M=N, N=N, K=hd, lda=3 * D, ldb=3 * D, ldc=N, batch=B * h, batch_inner=h,
             sA=(N * 3 * D, hd), sB=(N * 3 * D, hd), sC=(h * N * N, N * N), b_off=D, alpha=scale, c_f32=True)
        Pm = self._empty(B * h, N, N)
        check(lib.htrvt_softmax_rows(ptr(S), ptr(Pm), B * h * N, N, self.dti, None, 0, st), "softmax_rows")
        del S
        gemm(Pm, qkv, O, dtype=self.dtype, M=N, N=hd, K=N, lda=N, ldb=3 * D, ldc=D, b_layout=MNMAJOR, batch=B * h,
             batch_inner=h, sA=(h * N * N, N * N), sB=(N * 3 * D, hd), sC=(N * D, hd), b_off=2 * D)
        return Pm

    def _attention_bwd_unfused(self, qkv, Pm, dO, dqkv, B, N, D, h, hd, scale, st):
        """float32 path (and shapes the fused kernel does not serve): batched GEMMs + row-softmax backward over the saved P"""
        bstr = dict(batch=B * h, batch_inner=h)
        # dV = P^T dO
        gemm(Pm, dO, dqkv, dtype=self.dtype, M=N, N=hd, K=N, lda=N, ldb=D, ldc=3 * D, a_layout=MNMAJOR, b_layout=MNMAJOR,
             sA=(h * N * N, N * N), sB=(N * D, hd), sC=(N * 3 * D, hd), c_off=2 * D, **bstr)
        # dP = dO V^T
        dP = self._empty(B * h, N, N, dtype=torch.float32)
        gemm(dO, qkv, dP, dtype=self.dtype, M=N, N=N, K=hd, lda=D, ldb=3 * D, ldc=N, sA=(N * D, hd), sB=(N * 3 * D, hd),
             sC=(h * N * N, N * N), b_off=2 * D, c_f32=True, **bstr)
        dS = self._empty(B * h, N, N)
        check(lib.htrvt_softmax_bwd_rows(ptr(Pm), ptr(dP), ptr(dS), B * h * N, N, scale, self.dti, st), "softmax_bwd_rows")
        del dP
        # dQ = dS K ; dK = dS^T Q
        gemm(dS, qkv, dqkv, dtype=self.dtype, M=N, N=hd, K=N, lda=N, ldb=3 * D, ldc=3 * D, b_layout=MNMAJOR,
             sA=(h * N * N, N * N), sB=(N * 3 * D, hd), sC=(N * 3 * D, hd), b_off=D, c_off=0, **bstr)
        gemm(dS, qkv, dqkv, dtype=self.dtype, M=N, N=hd, K=N, lda=N, ldb=3 * D, ldc=3 * D, a_layout=MNMAJOR,
             b_layout=MNMAJOR, sA=(h * N * N, N * N), sB=(N * 3 * D, hd), sC=(N * 3 * D, hd), b_off=0, c_off=D, **bstr)
        del dS

    # ------------------------------------------------------------------ backward
    def backward(self, P, G, dy, after_encoder=None, after_layer3=None):
        """dy: float32 [B,N,C] = dLoss/dlogits.  Accumulates dLoss/dparam into G (dict name -> float32 tensor,
        same shapes as P; the caller zeroes it).  Uses the activations saved by forward(save=True)."""
        sv = self.saved
        assert sv is not None, "backward() needs forward(..., save=True)"
        s = self.s
        st = stream()
        B, N, D = sv["B"], sv["N"], s.D
        M = B * N
        C = s.nb_cls
        h, hd = s.heads, s.hd
        scale = hd ** -0.5
        dy = dy.contiguous()
        self._zarena_begin()
        self._pending_unpack = []
        self._bn_train = bool(sv["train"])
        assert dy.dtype == torch.float32 and dy.shape == (B, N, C)
        self._side_active = self.overlap_wgrad and not self.single_stream

        # sequence LN, head, final norm
        Cp = (C + 7) // 8 * 8           # class dim padded so that every 16-byte chunk is aligned
        draw = torch.zeros(M, Cp, dtype=self.dtype, device=self.dev)
        check(lib.htrvt_seq_whiten_bwd(ptr(dy), ptr(sv["y"]), ptr(sv["sstats"]), ptr(draw), B, N, C, Cp, self.dti, st),
              "seq_whiten_bwd")
        wh, wht = self._head_w(P["head.weight"])
        dxn = self.linear_dgrad(draw, wh, wht, plain=True)      # (split mode: the 80-class head stays on the float32 kernels)
        if Cp == C:
            self.linear_wgrad(draw, sv["xn"], G["head.weight"], G["head.bias"], plain=True)
        else:
            dwp, dbp = self._zeros(Cp, D), self._zeros(Cp)
            self.linear_wgrad(draw, sv["xn"], dwp, dbp, plain=True)
            self._join_side()
            check(lib.htrvt_rowsum_f32(ptr(dwp), 1, C * D, ptr(G["head.weight"]), st), "rowsum")
            check(lib.htrvt_rowsum_f32(ptr(dbp), 1, C, ptr(G["head.bias"]), st), "rowsum")
        dx = self.ln_bwd(dxn, sv["x_last"], sv["mn"], sv["rn"], P["norm.weight"], None, G["norm.weight"], G["norm.bias"])

        for e in reversed(sv["enc"]):
            p = e["p"]
            # MLP: x2 = x1 + fc2(gelu(fc1(ln2)))
            w2_, w2t = self._lin_w(p + ".mlp.fc2", P[p + ".mlp.fc2.weight"])
            dhpre = self.linear_dgrad(dx, w2_, w2t, act=2, preact=e["hpre"])
            self.linear_wgrad(dx, e["h"], G[p + ".mlp.fc2.weight"], G[p + ".mlp.fc2.bias"])
            w1_, w1t = self._lin_w(p + ".mlp.fc1", P[p + ".mlp.fc1.weight"])
            dln2 = self.linear_dgrad(dhpre, w1_, w1t)
            self.linear_wgrad(dhpre, e["ln2"], G[p + ".mlp.fc1.weight"], G[p + ".mlp.fc1.bias"])
            del dhpre
            dx1 = self.ln_bwd(dln2, e["x1"], e["m2"], e["r2"], P[p + ".norm2.weight"], dx, G[p + ".norm2.weight"],
                              G[p + ".norm2.bias"])
            # attention: x1 = x0 + proj(attn(ln1))
            wp, wpt = self._lin_w(p + ".attn.proj", P[p + ".attn.proj.weight"])
            dO = self.linear_dgrad(dx1, wp, wpt)
            self.linear_wgrad(dx1, e["O"], G[p + ".attn.proj.weight"], G[p + ".attn.proj.bias"])
            qkv, Pm = e["qkv"], e["P"]
            dqkv = self._empty(M, 3 * D)
            if Pm is None:      # fused forward: recomputing fused backward (dQ launch, then dK/dV launch)
                delta = self._empty(B * h, N, dtype=torch.float32)
                check(lib.htrvt_attn_bwd(ptr(qkv), None, ptr(e["O"]), ptr(dO), ptr(e["lse"]), ptr(delta), ptr(dqkv), None,
                                         B, N, h, hd, scale, self.dti, st), "attn_bwd")
            else:
                self._attention_bwd_unfused(qkv, Pm, dO, dqkv, B, N, D, h, hd, scale, st)
            wq, wqt = self._lin_w(p + ".attn.qkv", P[p + ".attn.qkv.weight"])
            dln1 = self.linear_dgrad(dqkv, wq, wqt)
            self.linear_wgrad(dqkv, e["ln1"], G[p + ".attn.qkv.weight"], G[p + ".attn.qkv.bias"])
            dx = self.ln_bwd(dln1, e["x0"], e["m1"], e["r1"], P[p + ".norm1.weight"], dx1, G[p + ".norm1.weight"],
                             G[p + ".norm1.bias"])

        if after_encoder is not None:   # every blocks.*, norm, head gradient is enqueued: DP bucket can go
            after_encoder(self._wgrad_stream())      # (the collective waits for the weight-gradient stream, not this one)

        # token assembly
        keep = sv["keep"]
        if keep is not None:
            ops.colsum(dx, M, D, D, G["mask_token"], dti=self.dti, keep=keep, keep_mod=N)
        Bq, Hc, Wc = sv["l3_shape"]
        dfeat = self._empty(Bq, Hc, Wc, D)
        check(lib.htrvt_pool_tokens_bwd(ptr(dx), ptr(sv["l3"]), ptr(keep), ptr(dfeat), B, Hc, N, D, self.dti, st),
              "pool_tokens_bwd")

        # residual stages.  bf16: the ReLU mask of a block's output and the BatchNorm-backward sums of its bn2 (and
        # downsample BN) are produced by the epilogue of the dgrad GEMM that creates that gradient; float32 (parity
        # path) keeps the separate reduction pass.
        blocks = sv["stem_blocks"]

        def can_fuse(g):   # served by the LDS-DMA kernel (bf16, > 128 rows per launch)
            return (self.fuse_bn_backward and self.dtype == torch.bfloat16 and
                    g.B * g.Hi * g.Wi // (g.sh * g.sw) > 128 and g.Ci % 8 == 0)

        def bnb_of(blk, C):
            """fused-sum request for the gradient flowing into `blk`'s output"""
            req = [(blk["cb"], blk["bn_b"][2], blk["bn_b"][3])]
            if blk["gd"] is not None:
                req.append((blk["cd"], blk["bn_d"][2], blk["bn_d"][3]))
            return req

        # gradient into the last block's output comes from the token kernel: unfused reduce for that one
        gm, parts = None, None
        dout = dfeat
        for bi in range(len(blocks) - 1, -1, -1):
            blk = blocks[bi]
            p = blk["p"]
            C = blk["cb"].shape[-1]
            if bi == len(blocks) - 3 and after_layer3 is not None:   # both layer-3 blocks (78 % of the stem's weights) are done
                self._flush_unpacks()
                after_layer3(self._wgrad_stream())
            if parts is None:   # dout is an unmasked gradient: classic path (mask + sums in one reduction pass)
                dcb, gm = self.bn_backward(dout, blk["out"], blk["cb"], p + ".bn2", P, G, blk["bn_b"][2], blk["bn_b"][3],
                                           want_g=True)
                parts_d = None
            else:               # dout is already g = dOut * (out > 0) and the sums exist
                gm = dout
                parts_d = parts[1] if len(parts) > 1 else None
            # downsample gradient as one more tap of the strided conv's class-(0,0) dgrad: d(conv1 out) and d(downsample
            # out) then live back to back in one allocation (the second gather source sits at a fixed offset from the first)
            fuse_ds = (self.fuse_downsample_dgrad and not self.split and blk["gd"] is not None and self._dgrad_by_class(blk["g1"]) and
                       2 * blk["ca"].numel() * blk["ca"].element_size() < 2 ** 31 - 64)     # A2 lies behind A inside ONE 2 GiB descriptor
            pair = self._empty(2, *blk["ca"].shape) if fuse_ds else None
            dca_out = pair[0] if fuse_ds else None
            dcd = None
            if parts is not None:
                if parts_d is not None and self.merge_bn_backward:
                    # bn2 and the downsample BN take the same gradient: one pass, gm read once (csrc/bwd.hip bn_bwd_apply2)
                    co_b = self.bn_backward_coef(parts[0][0], parts[0][1], blk["cb"], p + ".bn2", P, G, blk["bn_b"][2], blk["bn_b"][3])
                    co_d = self.bn_backward_coef(parts_d[0], parts_d[1], blk["cd"], p + ".downsample.1", P, G, blk["bn_d"][2], blk["bn_d"][3])
                    dcb = torch.empty_like(blk["cb"])
                    dcd = pair[1] if fuse_ds else torch.empty_like(blk["cd"])
                    check(lib.htrvt_bn_bwd_apply2(ptr(gm), ptr(blk["cb"]), ptr(co_b), ptr(dcb), ptr(blk["cd"]), ptr(co_d), ptr(dcd),
                                                  dcb.numel() // C, C, self.dti, stream()), "bn_bwd_apply2")
                else:
                    dcb, _ = self.bn_backward_finish(parts[0][0], parts[0][1], gm, None, blk["cb"], p + ".bn2", P, G,
                                                     blk["bn_b"][2], blk["bn_b"][3])
            _, wd2 = self._conv_w(p + ".conv2", P[p + ".conv2.weight"])
            self.conv_wgrad(dcb, blk["a1"], blk["g2"], G[p + ".conv2.weight"])
            if can_fuse(blk["g2"]):
                rows1 = self.dgrad_tiles(blk["g2"])
                part1 = self._empty(rows1, 2, C, dtype=torch.float32)
                # a1 = relu(bn1(ca)): the mask is recomputed from ca (read for the sums anyway) instead of reading a1
                rk = dict(relu_bn=(blk["bn_a"][0], blk["bn_a"][1])) if self.relu_mask_from_bn else dict(relu_src=blk["a1"])
                g1 = self.conv_dgrad(dcb, wd2, blk["g2"], bnb=[(blk["ca"], blk["bn_a"][2], blk["bn_a"][3], part1)], **rk)
                dca, _ = self.bn_backward_finish(part1, rows1, g1, None, blk["ca"], p + ".bn1", P, G, blk["bn_a"][2],
                                                 blk["bn_a"][3], out=dca_out)
                del g1
            else:
                da1 = self.conv_dgrad(dcb, wd2, blk["g2"])
                dca, _ = self.bn_backward(da1, blk["a1"], blk["ca"], p + ".bn1", P, G, blk["bn_a"][2], blk["bn_a"][3], out=dca_out)
                del da1
            del dcb
            _, wd1 = self._conv_w(p + ".conv1", P[p + ".conv1.weight"])
            self.conv_wgrad(dca, blk["x"], blk["g1"], G[p + ".conv1.weight"])
            # what the gradient of this block's INPUT feeds: the previous block's output (ReLU + bn2 [+ downsample BN])
            prev = blocks[bi - 1] if bi > 0 else None
            kw, parts = {}, None
            if prev is not None and can_fuse(blk["g1"]):
                Cp = prev["cb"].shape[-1]
                rows = self.dgrad_tiles(blk["g1"])
                req = bnb_of(prev, Cp)
                bufs = [self._empty(rows, 2, Cp, dtype=torch.float32) for _ in req]
                kw = dict(relu_src=prev["out"], bnb=[(x_, m_, r_, b_) for (x_, m_, r_), b_ in zip(req, bufs)])
                # the bit-mask form is compiled for (one sum set) and (residual + one or two sets)
                with_res = not (blk["gd"] is not None and fuse_ds)
                if prev.get("mask") is not None and (len(req) == 1 or with_res):
                    kw.update(relu_src=prev["mask"], relu_bits=True)
                parts = [(b_, rows) for b_ in bufs]
            if blk["gd"] is not None:
                dcd_out = pair[1] if fuse_ds else None
                if dcd is not None:
                    pass                      # came out of the joint pass with bn2 above
                elif parts_d is not None:
                    dcd, _ = self.bn_backward_finish(parts_d[0], parts_d[1], gm, None, blk["cd"], p + ".downsample.1", P, G,
                                                     blk["bn_d"][2], blk["bn_d"][3], out=dcd_out)
                else:
                    dcd, _ = self.bn_backward(gm, None, blk["cd"], p + ".downsample.1", P, G, blk["bn_d"][2], blk["bn_d"][3],
                                              out=dcd_out)
                _, wdd = self._conv_w(p + ".downsample.0", P[p + ".downsample.0.weight"])
                self.conv_wgrad(dcd, blk["x"], blk["gd"], G[p + ".downsample.0.weight"])
                if fuse_ds:
                    wdj = self._conv_w_joint_dgrad(p + ".conv1", P[p + ".conv1.weight"], p + ".downsample.0",
                                                   P[p + ".downsample.0.weight"])
                    dout = self.conv_dgrad(dca, wdj, blk["g1"], extra=dcd, **kw)
                else:
                    dres = self.conv_dgrad(dcd, wdd, blk["gd"])
                    dout = self.conv_dgrad(dca, wd1, blk["g1"], residual=dres, **kw)
            else:
                dout = self.conv_dgrad(dca, wd1, blk["g1"], residual=gm, **kw)

        # first maxpool + bn1 + conv1
        img, c1 = sv["img"], sv["c1"]
        u8 = 1 if img.dtype == torch.uint8 else 0
        sc, sf, mean, rstd = sv["bn1"]
        _, Hh, W, C1 = sv["c1_shape"]
        if self.fuse_conv1_backward and self._bn_train:
            # Cin = 1: dW1, dgamma, dbeta from per-channel sums over the pooled gradient (csrc/conv1_bwd.hip)
            partial = self._empty(lib.htrvt_conv1_bwd_rows(B, 2 * Hh), lib.htrvt_conv1_bwd_row_floats(C1), dtype=torch.float32)
            check(lib.htrvt_conv1_bwd(ptr(img), ptr(sv["stats"]), ptr(dout), ptr(sv["idx"]), ptr(P["patch_embed.conv1.weight"]),
                                      ptr(P["patch_embed.bn1.weight"]), ptr(mean), ptr(rstd), ptr(partial),
                                      ptr(G["patch_embed.conv1.weight"]), ptr(G["patch_embed.bn1.weight"]),
                                      ptr(G["patch_embed.bn1.bias"]), B, 2 * Hh, W, C1, self.dti, u8, st), "conv1_bwd")
        else:
            g = torch.empty_like(c1)
            check(lib.htrvt_maxpool_bwd(ptr(dout), ptr(sv["idx"]), ptr(c1), ptr(sc), ptr(sf), ptr(g), B, Hh, W, C1, self.dti, st),
                  "maxpool_bwd")
            dc1, _ = self.bn_backward(g, None, c1, "patch_embed.bn1", P, G, mean, rstd)
            del g
            nblk = lib.htrvt_conv1_wgrad_blocks(B, 2 * Hh)
            partial = self._empty(nblk, C1 * 9, dtype=torch.float32)
            check(lib.htrvt_conv1_wgrad(ptr(img), ptr(sv["stats"]), ptr(dc1), ptr(G["patch_embed.conv1.weight"]), ptr(partial),
                                        B, 2 * Hh, W, C1, self.dti, u8, st), "conv1_wgrad")
        self._flush_unpacks()
        self._join_side()
        self._side_active = False
        self._zarena_end()
        self._split_cache = []
        self._saved_planes = {}
        self.saved = None
