"""Attention flavours of the reference's variant models on the kernels of the hot path.

* model_window (/root/reference/model_window/model/HTR_VT.py:11-62,113-154): self-attention with a learned
  relative-position bias `table[(key - query) + P - 1, head]`, optionally restricted to 1-D windows of `window_size`
  tokens over the sequence rolled by `shift_size` (blocks 0 and 1 use windows of 16, shift 0 / 8).  Here: ONE dense
  additive bias [heads, N, N] -- table entries inside a window, a large negative number outside -- handed to the same
  fused attention kernels (bfloat16) or to the batched-GEMM + row-softmax path (float32); the gradient of the table is
  the kernels' dense d(score) summed over the batch and gathered back per table entry.  The bias is built, and its
  gradient gathered, by HIP kernels (csrc/variants.hip); a sequence that is not a multiple of the window (the
  reference zero-pads it and masks the padding keys) or of the kernels' 128-token tiles is handled by masked columns.
* model_sgm_* (/root/reference/model_sgm_2/model/sgm_head.py:118-127): the SGM head's single-head cross-attention
  softmax(Q K^T / sqrt(D)) K with K = V = the (normalised) visual tokens: batched htrvt_gemm + htrvt_softmax_rows over
  [B, L, D] queries and [B, N, D] tokens, forward and backward.

The index bookkeeping (which table entry each (query, key) pair uses) is host glue on tensors of heads * N^2 elements;
all arithmetic over activations runs in the HIP kernels."""
from __future__ import annotations

import torch

from ._lib import check, lib
from .ops import KMAJOR, MNMAJOR, colsum, dt, gemm, ptr, stream

MASKED = -1.0e30      # "outside the window": finite, so an all-masked key tile cannot produce inf - inf in the online softmax


def relative_position_index(N, num_patches, window_size=0, shift_size=0):
    """(index [N, N] int64 into the bias table, inside [N, N] bool) for queries i / keys j of the ORIGINAL sequence.
    Full attention (HTR_VT.py:27-31,45-46): index = (j - i) + P - 1.  Windowed (Block._attend, :113-154): the sequence is
    zero-padded to Np = a multiple of the window, token t sits at position (t - shift) mod Np of the rolled sequence,
    window = position // ws, slot = position % ws; a pair attends only inside one window and uses the slots' distance.
    The padding tokens are masked as keys (key_padding_mask, :47-56) and dropped as queries (:150-152), so among the N
    real tokens they only move the window boundaries -- which Np in the modulus reproduces."""
    t = torch.arange(N)
    if window_size <= 0:
        return (t[None, :] - t[:, None]) + num_patches - 1, torch.ones(N, N, dtype=torch.bool)
    Np = (N + window_size - 1) // window_size * window_size
    pos = (t - shift_size) % Np
    win, slot = pos // window_size, pos % window_size
    return (slot[None, :] - slot[:, None]) + num_patches - 1, win[None, :] == win[:, None]


def _padded_len(N, dtype, hd):
    """sequence length the attention kernels run at: a multiple of 128 where the fused bfloat16 kernels can serve it (the
    padding keys carry the bias -1e30, the padding queries are dropped), else a multiple of 8 (16-byte rows of the GEMMs)"""
    from ._lib import lib as _l
    n128 = (N + 127) // 128 * 128
    if dtype == torch.bfloat16 and _l.htrvt_attn_supported(n128, hd, dt(dtype)):
        return n128
    return (N + 7) // 8 * 8


class _RelPosBias(torch.autograd.Function):
    """table [(2P-1), heads] float32 (device) -> dense bias [heads, ld, ld]; both directions are HIP kernels
    (csrc/variants.hip): lookup + window mask forward, per-entry gather-sum backward (no atomics)"""

    @staticmethod
    def forward(ctx, table, N, num_patches, window_size, shift_size, ld):
        table = table.contiguous().float()
        heads = table.shape[1]
        bias = torch.empty(heads, ld, ld, dtype=torch.float32, device=table.device)
        check(lib.htrvt_relpos_bias_fwd(ptr(table), ptr(bias), N, num_patches, window_size, shift_size, heads, ld, stream()),
              "relpos_bias_fwd")
        ctx.geo = (N, num_patches, window_size, shift_size, heads, ld, tuple(table.shape))
        return bias

    @staticmethod
    def backward(ctx, dbias):
        N, P, ws, shift, heads, ld, tshape = ctx.geo
        dbias = dbias.contiguous().float()
        dtable = torch.empty(tshape, dtype=torch.float32, device=dbias.device)
        check(lib.htrvt_relpos_bias_bwd(ptr(dbias), ptr(dtable), N, P, ws, shift, heads, ld, stream()), "relpos_bias_bwd")
        return dtable, None, None, None, None, None


def relative_position_bias(table, N, num_patches, window_size=0, shift_size=0, ld=None):
    """dense float32 [heads, ld, ld] score bias from the learned table [(2 P - 1), heads] (ld >= N: the sequence length the
    attention kernels run at, see biased_self_attention; default N).  Differentiable in the table."""
    if not table.is_cuda:
        raise RuntimeError("htrvt_amd.variants needs device tensors on an MI355X (no CPU fallback); the index bookkeeping "
                           "alone is relative_position_index()")
    return _RelPosBias.apply(table, N, num_patches, window_size, shift_size, N if ld is None else ld)


class _BiasedSelfAttention(torch.autograd.Function):
    """qkv [B*N, 3*heads*hd] (the qkv Linear's output) + dense bias [heads, N, N] -> out [B*N, heads*hd]"""

    @staticmethod
    def forward(ctx, qkv, bias, B, N, heads):
        D = qkv.shape[1] // 3
        hd = D // heads
        scale = hd ** -0.5
        dti = dt(qkv.dtype)
        qkv, bias = qkv.contiguous(), bias.contiguous().float()
        out = torch.empty(B * N, D, dtype=qkv.dtype, device=qkv.device)
        st = stream()
        if lib.htrvt_attn_supported(N, hd, dti):
            lse = torch.empty(B * heads, N, dtype=torch.float32, device=qkv.device)
            check(lib.htrvt_attn_fwd(ptr(qkv), ptr(bias), ptr(out), ptr(lse), B, N, heads, hd, scale, dti, st), "attn_fwd")
            ctx.save_for_backward(qkv, bias, out, lse)
            ctx.fused = True
        else:
            S = torch.empty(B * heads, N, N, dtype=torch.float32, device=qkv.device)
            gemm(qkv, qkv, S, dtype=qkv.dtype, M=N, N=N, K=hd, lda=3 * D, ldb=3 * D, ldc=N, batch=B * heads, batch_inner=heads,
                 sA=(N * 3 * D, hd), sB=(N * 3 * D, hd), sC=(heads * N * N, N * N), b_off=D, alpha=scale, c_f32=True)
            P = torch.empty(B * heads, N, N, dtype=qkv.dtype, device=qkv.device)
            check(lib.htrvt_softmax_rows(ptr(S), ptr(P), B * heads * N, N, dti, ptr(bias), heads * N, st), "softmax_rows")
            gemm(P, qkv, out, dtype=qkv.dtype, M=N, N=hd, K=N, lda=N, ldb=3 * D, ldc=D, b_layout=MNMAJOR, batch=B * heads,
                 batch_inner=heads, sA=(heads * N * N, N * N), sB=(N * 3 * D, hd), sC=(N * D, hd), b_off=2 * D)
            ctx.save_for_backward(qkv, bias, out, P)
            ctx.fused = False
        ctx.dims = (B, N, heads, hd, D, scale, dti)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, N, h, hd, D, scale, dti = ctx.dims
        qkv, bias, out, aux = ctx.saved_tensors
        dout = dout.contiguous().to(qkv.dtype)
        dqkv = torch.empty_like(qkv)
        dbias = torch.zeros_like(bias)
        st = stream()
        if ctx.fused:
            delta = torch.empty(B * h, N, dtype=torch.float32, device=qkv.device)
            check(lib.htrvt_attn_bwd(ptr(qkv), ptr(bias), ptr(out), ptr(dout), ptr(aux), ptr(delta), ptr(dqkv), ptr(dbias), B, N, h,
                                     hd, scale, dti, st), "attn_bwd")
            return dqkv, dbias, None, None, None
        P = aux
        bstr = dict(batch=B * h, batch_inner=h)
        gemm(P, dout, dqkv, dtype=qkv.dtype, M=N, N=hd, K=N, lda=N, ldb=D, ldc=3 * D, a_layout=MNMAJOR, b_layout=MNMAJOR,
             sA=(h * N * N, N * N), sB=(N * D, hd), sC=(N * 3 * D, hd), c_off=2 * D, **bstr)                     # dV = P^T dO
        dP = torch.empty(B * h, N, N, dtype=torch.float32, device=qkv.device)
        gemm(dout, qkv, dP, dtype=qkv.dtype, M=N, N=N, K=hd, lda=D, ldb=3 * D, ldc=N, sA=(N * D, hd), sB=(N * 3 * D, hd),
             sC=(h * N * N, N * N), b_off=2 * D, c_f32=True, **bstr)                                           # dP = dO V^T
        dS = torch.empty(B * h, N, N, dtype=qkv.dtype, device=qkv.device)          # d(score), the scale goes into the GEMMs
        check(lib.htrvt_softmax_bwd_rows(ptr(P), ptr(dP), ptr(dS), B * h * N, N, 1.0, dti, st), "softmax_bwd_rows")
        colsum(dS, B, h * N * N, h * N * N, dbias, dti=dti)                         # d(bias) = sum over the batch
        gemm(dS, qkv, dqkv, dtype=qkv.dtype, M=N, N=hd, K=N, lda=N, ldb=3 * D, ldc=3 * D, b_layout=MNMAJOR, alpha=scale,
             sA=(h * N * N, N * N), sB=(N * 3 * D, hd), sC=(N * 3 * D, hd), b_off=D, c_off=0, **bstr)          # dQ = dS K
        gemm(dS, qkv, dqkv, dtype=qkv.dtype, M=N, N=hd, K=N, lda=N, ldb=3 * D, ldc=3 * D, a_layout=MNMAJOR, b_layout=MNMAJOR,
             alpha=scale, sA=(h * N * N, N * N), sB=(N * 3 * D, hd), sC=(N * 3 * D, hd), b_off=0, c_off=D, **bstr)   # dK = dS^T Q
        return dqkv, dbias, None, None, None


def biased_self_attention(qkv, bias, B, N, heads):
    """softmax(q k^T * hd^-0.5 + bias) v over qkv [B*N, 3*heads*hd] (layout [B,N,3,heads,hd]); bias [heads, ld, ld] float32
    with ld >= N (see relative_position_bias; columns >= N must hold -1e30).  bfloat16 with hd in {32, 64, 128}: the fused
    kernels, the sequence zero-padded to a multiple of 128 if it is not one (masked keys, dropped queries); otherwise
    (float32 parity path) batched GEMMs + row softmax at a multiple of 8.  Differentiable in qkv and bias."""
    if not qkv.is_cuda:
        raise RuntimeError("htrvt_amd.variants needs device tensors on an MI355X (no CPU fallback)")
    D3 = qkv.shape[1]
    hd = D3 // 3 // heads
    Np = _padded_len(N, qkv.dtype, hd)
    ld = bias.shape[-1]
    if ld != Np:
        if ld != N:
            raise ValueError(f"bias is [{heads}, {ld}, {ld}]: expected ld = {N} or the padded length {Np}")
        padded = torch.full((heads, Np, Np), MASKED, dtype=torch.float32, device=bias.device)
        padded[:, :N, :N] = bias            # (glue: a strided copy; use relative_position_bias(..., ld=padded_len) to avoid it)
        padded[:, N:, 0] = 0.0
        bias = padded
    if Np == N:
        return _BiasedSelfAttention.apply(qkv, bias, B, N, heads)
    qp = torch.zeros(B, Np, D3, dtype=qkv.dtype, device=qkv.device)
    qp[:, :N] = qkv.view(B, N, D3)
    out = _BiasedSelfAttention.apply(qp.view(B * Np, D3), bias, B, Np, heads)
    return out.view(B, Np, -1)[:, :N].reshape(B * N, -1)


def padded_len(N, dtype, head_dim):
    """the `ld` to build the bias at for biased_self_attention"""
    return _padded_len(N, dtype, head_dim)


class _CrossAttention(torch.autograd.Function):
    """SGMHead._cross_attend (sgm_head.py:118-127): out = softmax(Q K^T / sqrt(D)) K, Q [B,L,D], K = V [B,N,D]"""

    @staticmethod
    def forward(ctx, Q, KV):
        B, L, D = Q.shape
        N = KV.shape[1]
        dtype, dti = Q.dtype, dt(Q.dtype)
        Q, KV = Q.contiguous(), KV.contiguous()
        scale = D ** -0.5
        S = torch.empty(B, L, N, dtype=torch.float32, device=Q.device)
        gemm(Q, KV, S, dtype=dtype, M=L, N=N, K=D, lda=D, ldb=D, ldc=N, batch=B, sA=(L * D, 0), sB=(N * D, 0), sC=(L * N, 0),
             alpha=scale, c_f32=True)
        P = torch.empty(B, L, N, dtype=dtype, device=Q.device)
        check(lib.htrvt_softmax_rows(ptr(S), ptr(P), B * L, N, dti, None, 0, stream()), "softmax_rows")
        out = torch.empty(B, L, D, dtype=dtype, device=Q.device)
        gemm(P, KV, out, dtype=dtype, M=L, N=D, K=N, lda=N, ldb=D, ldc=D, b_layout=MNMAJOR, batch=B, sA=(L * N, 0),
             sB=(N * D, 0), sC=(L * D, 0))
        ctx.save_for_backward(Q, KV, P)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        Q, KV, P = ctx.saved_tensors
        B, L, D = Q.shape
        N = KV.shape[1]
        dtype, dti, scale = Q.dtype, dt(Q.dtype), ctx.scale
        dout = dout.contiguous().to(dtype)
        bb = dict(batch=B)
        dP = torch.empty(B, L, N, dtype=torch.float32, device=Q.device)
        gemm(dout, KV, dP, dtype=dtype, M=L, N=N, K=D, lda=D, ldb=D, ldc=N, sA=(L * D, 0), sB=(N * D, 0), sC=(L * N, 0),
             c_f32=True, **bb)                                                        # dP = dO V^T
        dS = torch.empty(B, L, N, dtype=dtype, device=Q.device)
        check(lib.htrvt_softmax_bwd_rows(ptr(P), ptr(dP), ptr(dS), B * L, N, scale, dti, stream()), "softmax_bwd_rows")
        dQ = torch.empty_like(Q)
        gemm(dS, KV, dQ, dtype=dtype, M=L, N=D, K=N, lda=N, ldb=D, ldc=D, b_layout=MNMAJOR, sA=(L * N, 0), sB=(N * D, 0),
             sC=(L * D, 0), **bb)                                                     # dQ = dS K
        dKV = torch.empty_like(KV)
        gemm(P, dout, dKV, dtype=dtype, M=N, N=D, K=L, lda=N, ldb=D, ldc=D, a_layout=MNMAJOR, b_layout=MNMAJOR,
             sA=(L * N, 0), sB=(L * D, 0), sC=(N * D, 0), **bb)                       # through V: P^T dO
        gemm(dS, Q, dKV, dtype=dtype, M=N, N=D, K=L, lda=N, ldb=D, ldc=D, a_layout=MNMAJOR, b_layout=MNMAJOR,
             sA=(L * N, 0), sB=(L * D, 0), sC=(N * D, 0), residual=dKV, **bb)         # + through K: dS^T Q
        return dQ, dKV


def cross_attention(Q, KV):
    """single-head cross-attention of the SGM head: Q [B, L, D] text queries, KV [B, N, D] visual tokens (K = V).
    N and D multiples of 8 (bfloat16) / 4 (float32); any L."""
    if not (Q.is_cuda and KV.is_cuda):
        raise RuntimeError("htrvt_amd.variants needs device tensors on an MI355X (no CPU fallback)")
    return _CrossAttention.apply(Q, KV)
