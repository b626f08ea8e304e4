/* htrvt.h -- C ABI of libhtrvt_hip.so, the MI355X (gfx950) kernels behind the
 * HTR-VT forward / training hot path.
 *
 * The reference (0xk0ry/HTR-VT) has NO native layer and no FFI: every FLOP of
 * its hot path is an ATen call issued from model_v1/model/HTR_VT.py,
 * model_v1/model/resnet18.py and model_v1/train.py:21-30.  Each entry point
 * below therefore cites the ATen call site(s) it replaces.  The Python drop-in
 * boundary (create_model / forward) lives in htr-vt_amd/model/HTR_VT.py and
 * binds these symbols with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by
 *     the caller (PyTorch); the library allocates nothing and keeps no state.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it.
 *   - return 0 on success, negative on error; htrvt_last_error() describes it.
 *   - dtype: 0 = float32, 1 = bfloat16 (storage type of activations / packed
 *     weights).  Statistics, biases, gains, master weights are always float32.
 *   - activations are NHWC ([B,H,W,C]); tokens are [B,N,D] row-major.
 */
#ifndef HTRVT_H
#define HTRVT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HTRVT_F32 0
#define HTRVT_BF16 1

#define HTRVT_KMAJOR 0  /* element (row,k) at base + row*ld + k   */
#define HTRVT_MNMAJOR 1 /* element (row,k) at base + k*ld + row   */

#define HTRVT_GATHER_NONE 0
#define HTRVT_GATHER_CONV_FWD 1   /* A rows = output pixels, K = taps*Cpad(Ci)  */
#define HTRVT_GATHER_CONV_DGRAD 2 /* A rows = input pixels,  K = taps*Cpad(Co)  */
#define HTRVT_GATHER_CONV_WGRAD 3 /* A = x gathered (MN-major, k = output pixel, M = taps*Cpad(Ci)), B = dy, N = Co */

int htrvt_version(void);
const char* htrvt_last_error(void);
/* symbol of the MFMA kernel the last htrvt_gemm call of this thread launched, spelled as rocprofv3 prints it
 * (e.g. "gemm_dma_kernel<256, 192, 0, 0, 2, 1, 2>"): lets bench.py name its roofline kernel by the profiler's symbol */
const char* htrvt_last_kernel(void);

/* One MFMA GEMM   C[m][n] = alpha * sum_k A(m,k) * B(n,k)  (+ epilogue).
 * Replaces: nn.Linear / torch.matmul (HTR_VT.py:22,29-37,170; timm Mlp fc1/fc2),
 * F.conv2d 3x3 / 1x1 (resnet18.py:6-7,26-31,59-63) as implicit GEMM over NHWC,
 * and their autograd backward (train.py:123). */
typedef struct HtrvtGemmDesc {
  int32_t dtype;            /* element type of A and B (and of C unless c_f32)      */
  int32_t a_layout, b_layout;
  int32_t gather;
  int32_t M, N, K;
  int64_t lda, ldb, ldc;    /* in elements                                          */
  int32_t batch, batch_inner; /* batch index z -> (z / batch_inner, z % batch_inner) */
  int64_t sA_o, sA_i, sB_o, sB_i, sC_o, sC_i; /* batch strides in elements          */
  int32_t split_k;          /* >1: grid.z splits K, result accumulated (needs accumulate=1, c_f32=1) */
  /* split_k > 1 only.  NULL: the K ranges meet in C through float atomics (summation order = arrival order).
   * Non-NULL: float32 scratch of split_k*M*N elements; every K range stores its own [M][N] slab with plain stores and a
   * second launch adds the slabs into C in ascending K order -- bitwise reproducible (the float32 parity path). */
  float* splitk_ws;
  /* conv geometry (gather != 0) */
  int32_t nB, Hi, Wi, Ci, Ho, Wo, Co, kh, kw, sh, sw, ph, pw, Cpad;
  /* HTRVT_GATHER_CONV_DGRAD of a strided conv, one launch per input-pixel parity class (bfloat16 only):
   * cls_h/cls_w in [0,sh)/[0,sw) select the pixels hi%sh==cls_h, wi%sw==cls_w (M = their count, row-major over
   * (b, hi/sh, wi/sw)); K = (#taps that can reach this class) * Cpad (may be 0).  cls_h = -1: all pixels/taps.
   * cls_h = cls_w = -2: ALL parity classes of a 3x3 pad-1 stride-(2,1) / (2,2) convolution in ONE launch on halo-staged
   * tiles (csrc/gemm_halo_impl.h: gemm_halo_s2_kernel; resnet18.py:26 with stride from :59-63): M = nB*Hi*Wi, K = 9 * Cpad
   * (10 * Cpad with A2, which then rides with the class-(0,0) tiles), the epilogue / side inputs address C rows as NHWC
   * pixels, bnb_partial gets one row per (class, 256-pixel segment) tile in grid order.  Served only where
   * htrvt_gemm_dgrad_merged_tiles() says so (Hi even, Wo a multiple of 256, operands inside 2 GiB). */
  int32_t cls_h, cls_w;
  /* Parity-class dgrad of class (0, 0) only, or NULL: A2 = the gradient of the block's 1x1 downsample convolution output
   * (resnet18.py:59-63: same stride, padding 0, same [B,Ho,Wo,Co] shape as A, allocated BEHIND A within 2 GiB).  Its
   * input gradient lands on exactly the class-(0,0) pixels, through the pixel the centre tap of the 3x3 reaches, so it is
   * contracted as ONE MORE TAP: K = (#taps of the class + 1) * Cpad, and B holds kh*kw + 1 taps per row (ldb =
   * (kh*kw + 1) * Cpad), the last one the packed 1x1 weight.  Replaces a separate 1x1 dgrad (one launch per class, three
   * of them without a single tap) plus a residual round trip of the whole input gradient. */
  const void* A2;
  /* epilogue */
  float alpha;
  int32_t act;              /* 0 none, 1 exact-erf GELU, 2 multiply by GELU'(preact), 3 ReLU applied last (after residual) */
  int32_t c_f32;            /* 1: C (and residual) are float32 regardless of dtype  */
  int32_t accumulate;       /* 1: atomicAdd into float32 C                           */
  int32_t tile;             /* 0 auto; else BM*1000+BN of a built instantiation     */
  const float* bias;        /* [N] or NULL                                           */
  const float* colscale;    /* [N] or NULL: C = alpha*acc*colscale[n] + bias[n] (eval-mode BatchNorm folded into the conv) */
  void* preact;             /* same shape/type as C: value before `act`, or NULL     */
  const void* residual;     /* same shape/type as C, added last, or NULL             */
  float* colstats;          /* [ceil(M/BM)][2][N]: per-M-tile column sum / sum of squares of the
                               float accumulators (before bias), or NULL           */
  /* bfloat16 conv-dgrad outputs only (LDS-DMA kernel with loader waves, N tile 192/128/64):
   * relu_src: same shape/type as C; C = (relu_src > 0) ? value : 0, applied after `residual` (backward of ReLU).
   * bnb_*[t], t = 0,1: BatchNorm-backward sums of the (masked) gradient written to C against up to two raw conv
   * outputs bnb_x[t] (same shape as C): bnb_partial[t][bnb_tile0 + m_tile][2][N] = { sum g, sum g*(x-mean)*rstd }.
   * bnb_partial[0] == NULL disables it.
   * relu_scale / relu_shift (float32 [N], both or neither; with bnb_partial[0] set, bnb_partial[1], residual and relu_src
   * NULL): the ReLU whose backward this is was applied to BatchNorm(bnb_x[0]) -- conv -> bn1 -> relu -> conv2 inside a
   * BasicBlock, resnet18.py:27-31 -- so its mask is recomputed from the BatchNorm input the epilogue reads anyway:
   * C = (bnb_x[0] * relu_scale[n] + relu_shift[n] > 0) ? value : 0, the same fused multiply-add htrvt_bn_apply formed in
   * the forward.  Saves the read of the activation tensor (403 MB per layer-1 launch). */
  const void* relu_src;
  const void* bnb_x[2];
  const float* bnb_mean[2];
  const float* bnb_rstd[2];
  float* bnb_partial[2];
  int32_t bnb_tile0;
  const float* relu_scale;
  const float* relu_shift;
  /* 1: relu_src is a BIT mask instead of the activation itself -- bit (m * ldc + n) of the byte array, i.e. one byte per
   * 8 consecutive columns of a row, set where the ReLU's output was positive (htrvt_bn_apply_mask writes it in the forward
   * pass): 1/16 of the side-input bytes of the fused dgrad epilogue (403 -> 25 MB per layer-1 launch) */
  int32_t relu_bits;
  const void* A;
  const void* B;
  void* C;
} HtrvtGemmDesc;

int htrvt_gemm(const HtrvtGemmDesc* d, void* stream);
/* rows of colstats (= number of M tiles) the call above will write for this desc */
int htrvt_gemm_num_mtiles(const HtrvtGemmDesc* d);
/* Weight-gradient launches (gather = HTRVT_GATHER_CONV_WGRAD; or gather = 0 with both operands MN-major and float32 output: a
 * Linear weight gradient, 256 x 256 tiles when the MN-major 8-phase kernel serves it): which tiling htrvt_gemm would use.  Returns the number of
 * workgroups per K range and the tile extents when the halo-staged kernel (3x3, W stride 1, row length a multiple of 64,
 * Cpad a multiple of 64, float32 output, operands inside 2 GiB) serves the descriptor, 0 when the generic kernels do: the
 * caller's split-K choice must count the workgroups of the kernel that actually runs. */
int htrvt_gemm_wgrad_tiling(const HtrvtGemmDesc* d, int* tile_rows, int* tile_cols);
/* Merged strided dgrad (cls_h = cls_w = -2 above): the number of M tiles (= rows every bnb_partial buffer needs) when
 * htrvt_gemm serves the descriptor in that form, 0 when the caller has to launch one parity class at a time. */
int htrvt_gemm_dgrad_merged_tiles(const HtrvtGemmDesc* d);

/* ---- stem helpers (resnet18.py:42-84, HTR_VT.py:134-136,224-227) ------------- */
/* per-image mean / rstd of the raw image (param-free LayerNorm, eps 1e-5): stats[b] = {mean, rstd}.
 * img_u8 (here and in the conv1 entry points): 0 = float32 pixels; 1 = uint8 pixels taken as value / 255, i.e. the
 * torchvision ToTensor hand-off of the reference's data pipeline (data/dataset.py) fused into the first reads of the
 * image -- a quarter of the host->device bytes. */
int htrvt_img_stats(const void* img, float* stats, int B, int HW, float eps, int img_u8, void* stream);
/* conv1: whiten + 3x3 conv Cin=1 stride (2,1) pad 1 -> NHWC [B,H/2,W,C] + BN partial sums
 * colstats[B*H/2][2][C] (one row per output image row). w: [C][9] float32. */
int htrvt_conv1_fwd(const void* img, const float* stats, const float* w, void* out, float* colstats,
                    int B, int H, int W, int C, int dtype, int img_u8, void* stream);
/* Fused stem forward (resnet18.py:74-77 with HTR_VT.py:224): conv1 (one input channel) -> BatchNorm -> ReLU ->
 * max_pool2d(3, stride (2,1), pad 1) straight from the image; the conv1 tensor is never written.
 * htrvt_stem_stats: per-channel (sum, sum of squares) of the conv1 output computed from 54 second moments of the image
 *   (batch statistics of a train-mode BatchNorm without the tensor) -> colstats float32 [2][C], one partial row for
 *   htrvt_bn_finalize(rows = 1, count = B * H/2 * W); partial: float32 [htrvt_stem_stats_rows(B, H)][64] workspace.
 * htrvt_stem_fwd: y [B][Hp][W][C] in dtype (Hp = (H/2 - 1)/2 + 1), idx uint8 (arg-max 3*row + column, 15 = ReLU
 *   closed; NULL to skip) -- same values and arg-max rule as htrvt_conv1_fwd + htrvt_bn_relu_maxpool. */
int htrvt_stem_stats_rows(int B, int H);
int htrvt_stem_stats(const void* img, const float* stats, const float* w, float* partial, float* colstats, int B, int H,
                     int W, int C, int img_u8, void* stream);
int htrvt_stem_fwd(const void* img, const float* stats, const float* w, const float* scale, const float* shift, void* y,
                   uint8_t* idx, int B, int H, int W, int C, int dtype, int img_u8, void* stream);
/* BN statistics from partial sums: train mode.  partial [rows][2][C]; count = #elements per channel.
 * Writes scale = gamma*rstd, shift = beta - mean*scale, saves mean/rstd, updates running stats
 * (momentum, unbiased running_var) when running_mean != NULL and adds 1 to *num_batches_tracked (int64, may be NULL). */
int htrvt_bn_finalize(const float* partial, int rows, int C, float count, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                      float* scale, float* shift, float* save_mean, float* save_rstd, void* stream);
/* eval mode: scale/shift from running stats; rstd (may be NULL) = 1/sqrt(running_var + eps) for an eval-mode backward */
int htrvt_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                         float eps, float* scale, float* shift, float* rstd, int C, void* stream);
/* y = [relu]( x*scale+shift [+ (res*rscale+rshift | res)] ), NHWC, any number of pixels */
int htrvt_bn_apply(const void* x, const float* scale, const float* shift, const void* res, const float* rscale,
                   const float* rshift, void* y, int64_t npix, int C, int relu, int dtype, void* stream);
/* the same pass, additionally writing the ReLU bit mask of y: mask[i >> 3] bit (i & 7) = (y[i] > 0) over the flattened
 * [npix][C] output (C a multiple of 8; bfloat16 only) -- the `relu_bits` side input of the dgrad launch that takes the
 * backward of this ReLU (HtrvtGemmDesc.relu_bits), so that the backward reads 1 bit instead of 16 per element */
int htrvt_bn_apply_mask(const void* x, const float* scale, const float* shift, const void* res, const float* rscale,
                        const float* rshift, void* y, uint8_t* mask, int64_t npix, int C, int relu, int dtype, void* stream);
/* y = maxpool3x3 stride (2,1) pad 1 ( relu(x*scale+shift) ), NHWC; scale==NULL: plain maxpool of x.
 * idx (uint8 per output element, window position 0..8 of the FIRST maximum in scan order, as ATen) or NULL */
int htrvt_bn_relu_maxpool(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx,
                          int B, int H, int W, int C, int dtype, void* stream);
/* tokens[b][n][:] = (keep[n] ? maxpool(x)[b,0,n,:] : mask_token) + pos[n][:];  x NHWC [B,H(<=3),N,D] */
int htrvt_pool_tokens(const void* x, const float* keep, const float* mask_token, const float* pos, void* tok,
                      int B, int H, int N, int D, int dtype, void* stream);

/* ---- encoder helpers (HTR_VT.py:27-39,68-83,169-170,236-239) ------------------ */
int htrvt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                        int64_t rows, int D, float eps, int dtype, void* stream);
/* row softmax of float32 scores [rows][n] -> probabilities of type dtype in `p`.  bias (may be NULL): float32
 * [bias_rows][n] added to the scores first, score row r taking bias row r % bias_rows (rows ordered [batch][head][query]
 * with bias_rows = heads * N: the relative-position / window bias of SURVEY 8(f-4)). */
int htrvt_softmax_rows(const float* s, void* p, int64_t rows, int n, int dtype, const float* bias, int64_t bias_rows,
                       void* stream);
/* Fused multi-head self-attention, bfloat16 (HTR_VT.py:27-36: softmax(q k^T * scale) v and its autograd backward).
 * qkv [B*N][3][heads][hd] (the qkv Linear's output), out / dout [B*N][heads][hd], dqkv like qkv; scores and
 * probabilities stay on chip.  lse2 [B*heads][N] float32 (may be NULL in the forward when no backward follows):
 * log2 of the softmax denominator in the scaled base-2 domain, P = exp2(S * scale * log2(e) - lse2).
 * delta [B*heads][N] float32: scratch of the backward (rowsum(dout * out), written by its first launch).
 * htrvt_attn_supported: any N >= 32 (partial last query / key tiles are masked inside the kernels), hd in {32, 64, 128},
 * dtype bfloat16 (others: the htrvt_gemm + htrvt_softmax_rows path).
 * bias (may be NULL): float32 [heads][N][N] added to the scaled scores before the softmax -- the variant blocks of
 * relative-position bias table gathered per (query, key) and, for 1-D windowed / shifted attention, a
 * large negative number (-1e30, not -inf) outside the query's window (model_window/model/HTR_VT.py:23-56,113-154);
 * dbias (NULL exactly when bias is NULL): float32 [heads][N][N] += sum over the batch of d(loss)/d(score) (float atomics). */
int htrvt_attn_supported(int N, int hd, int dtype);
int htrvt_attn_fwd(const void* qkv, const float* bias, void* out, float* lse2, int B, int N, int heads, int hd, float scale,
                   int dtype, void* stream);
int htrvt_attn_bwd(const void* qkv, const float* bias, const void* out, const void* dout, const float* lse2, float* delta,
                   void* dqkv, float* dbias, int B, int N, int heads, int hd, float scale, int dtype, void* stream);
/* param-free LN over all N*C logits of a sample (HTR_VT.py:136,239): in dtype -> out float32 */
int htrvt_seq_whiten_fwd(const void* x, float* y, float* stats, int B, int NC, float eps, int dtype, void* stream);

/* ---- fused log-softmax + CTC (train.py:21-30; ATen ctc_loss with cuDNN off) ---- */
/* logits [B][T][C] float32; targets concatenated int32; tgt_len/tgt_off [B] int32.
 * nll[b] (0 if infeasible, zero_infinity); grad[b][t][c] = grad_scale * d(mean_b nll)/dlogits (NULL to skip;
 * grad_scale = 1/world_size folds the data-parallel gradient average into the loss).
 * workspace: float32, htrvt_ctc_workspace_floats(B, T, max_target_len) elements (alpha, beta, frame log-sum-exp).
 * Targets of up to 127 labels: frame log-sum-exp, the alpha and beta recursions side by side (one wave each per
 * sample), then the gradient of all (b, t) rows in parallel; longer targets: one workgroup per sample. */
size_t htrvt_ctc_workspace_floats(int B, int T, int max_target_len);
int htrvt_ctc_loss(const float* logits, const int32_t* targets, const int32_t* tgt_len, const int32_t* tgt_off,
                   float* nll, float* grad, float* workspace, int B, int T, int C, int max_target_len,
                   float grad_scale, void* stream);

/* ---- backward of the path (autograd of HTR_VT.py:222-241, train.py:123) -------- */
/* Every parameter-gradient output below is float32 and is ACCUMULATED (+=). */
/* dx = rstd*(dy - mean(dy) - y*mean(dy*y)) per sample over its N*C logits; dx has type dtype and row
 * stride ldo >= C (pad columns are left untouched) */
int htrvt_seq_whiten_bwd(const float* dy, const float* y, const float* stats, void* dx, int B, int N, int C, int ldo,
                         int dtype, void* stream);
/* LayerNorm backward: dx = LN'(dy) [+ dres]; partial[blocks][2][D] = per-block {dgamma, dbeta} */
int htrvt_layernorm_bwd_blocks(int64_t rows);
int htrvt_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                        const void* dres, void* dx, float* partial, int64_t rows, int D, int dtype, void* stream);
/* dS = scale * P * (dP - rowsum(dP*P)) ; P, dS of type dtype, dP float32 */
int htrvt_softmax_bwd_rows(const void* p, const float* dp, void* ds, int64_t rows, int n, float scale, int dtype,
                           void* stream);
/* out[c] += sum_r x[r*ld + c]; rows with keep[r % keep_mod] != 0 are skipped when keep != NULL.  Reproducible
 * two-stage sum (no float atomics): workspace = htrvt_colsum_workspace_floats(rows, cols) floats (may be 0 -> NULL). */
size_t htrvt_colsum_workspace_floats(int64_t rows, int cols);
int htrvt_colsum(const void* x, int64_t rows, int cols, int64_t ld, float* out, const float* keep, int keep_mod,
                 int dtype, float* workspace, void* stream);
int htrvt_rowsum_f32(const float* partial, int rows, int cols, float* out, void* stream);
/* train-mode BatchNorm backward (resnet18.py:27-37 under autograd): g = dy * (yact > 0) when yact != NULL */
int htrvt_bn_bwd_blocks(int64_t npix);
int htrvt_bn_bwd_reduce(const void* dy, const void* yact, const void* x, const float* mean, const float* rstd,
                        float* partial, int64_t npix, int C, int dtype, void* stream);
/* count = elements per channel (train mode: batch statistics).  count <= 0: eval-mode BatchNorm, mean / rstd are the
 * running statistics (constants): same dgamma / dbeta, dx = gamma * rstd * g. */
int htrvt_bn_bwd_finalize(const float* partial, int rows, int C, float count, const float* gamma, const float* mean,
                          const float* rstd, float* dgamma, float* dbeta, float* coef /* [3][C] */, void* stream);
int htrvt_bn_bwd_apply(const void* dy, const void* yact, const void* x, const float* coef, void* dx, void* gout,
                       int64_t npix, int C, int dtype, void* stream);
/* two BatchNorms fed by the same (already masked) gradient g -- bn2 and the downsample BN of a stage's first block,
 * resnet18.py:33-37: dx1 = BN1-backward(g, x1), dx2 = BN2-backward(g, x2) with one read of g */
int htrvt_bn_bwd_apply2(const void* g, const void* x1, const float* coef1, void* dx1, const void* x2, const float* coef2,
                        void* dx2, int64_t npix, int C, int dtype, void* stream);
/* backward of htrvt_bn_relu_maxpool: g = d(bn output, ReLU-masked), x = raw conv output [B,H,W,C] */
int htrvt_maxpool_bwd(const void* dpool, const uint8_t* idx, const void* x, const float* scale, const float* shift,
                      void* g, int B, int H, int W, int C, int dtype, void* stream);
/* backward of htrvt_pool_tokens w.r.t. x (masked tokens pass no gradient) */
int htrvt_pool_tokens_bwd(const void* dtok, const void* x, const float* keep, void* dx, int B, int H, int N, int D,
                          int dtype, void* stream);
/* dW[C][9] += sum_pix dY * whitened-input tap (autograd of htrvt_conv1_fwd); partial: [blocks][C*9] float32 scratch */
int htrvt_conv1_wgrad_blocks(int B, int H);
int htrvt_conv1_wgrad(const void* img, const float* stats, const void* dy, float* dw, float* partial,
                      int B, int H, int W, int C, int dtype, int img_u8, void* stream);

/* Backward of conv1 -> BatchNorm(train) -> ReLU -> max_pool2d(3,(2,1),1) (resnet18.py:74-77 under autograd) with
 * respect to conv1.weight [C][9], bn1.weight and bn1.bias, in one pass over the pooled gradient dpool [B,Hp,W,C] and
 * the arg-max bytes written by htrvt_bn_relu_maxpool (Cin = 1: the chain rule reduces to per-channel sums, see
 * csrc/conv1_bwd.hip).  img [B,H,W] float32, stats = htrvt_img_stats output, w = conv1.weight, mean/rstd = the batch
 * statistics saved by htrvt_bn_finalize.  partial: float32 scratch of htrvt_conv1_bwd_rows(B,H) rows x
 * htrvt_conv1_bwd_row_floats(C) floats.  dw, dgamma, dbeta accumulate. */
int htrvt_conv1_bwd_rows(int B, int H);
int htrvt_conv1_bwd_row_floats(int C);
int htrvt_conv1_bwd(const void* img, const float* stats, const void* dpool, const uint8_t* idx, const float* w,
                    const float* gamma, const float* mean, const float* rstd, float* partial, float* dw, float* dgamma,
                    float* dbeta, int B, int H, int W, int C, int dtype, int img_u8, void* stream);

/* ---- variant blocks: relative-position bias of the window-attention fork -------------------------
 * model_window/model/HTR_VT.py:23-31,45-46 (table [(2P-1)][heads] + index) and :113-154 (Block._attend: pad to a multiple
 * of the window, cyclic shift, 1-D windows, key_padding_mask) as one dense additive score bias [heads][ld][ld] for
 * htrvt_attn_fwd / htrvt_softmax_rows: table entry inside a window, -1e30 outside and in the columns N..ld-1 (sequence
 * padded for the kernels; rows N..ld-1 are don't-care queries).  window <= 0: full attention.
 * _bwd: dtable[(2P-1)][heads] = gather-sum of dbias over the (query, key) pairs of each entry, fixed order, overwrites. */
int htrvt_relpos_bias_fwd(const float* table, float* bias, int N, int num_patches, int window, int shift, int heads, int ld,
                          void* stream);
int htrvt_relpos_bias_bwd(const float* dbias, float* dtable, int N, int num_patches, int window, int shift, int heads, int ld,
                          void* stream);

/* ---- weight layout helpers ----------------------------------------------------- */
/* w [Co][Ci][taps] float32 -> fwd [Co][taps][cpad_in], dgrad [Ci][taps][cpad_out] (may be NULL); pads untouched */
int htrvt_pack_conv_weight(const float* w, void* fwd, void* dgrad, int Co, int Ci, int taps, int cpad_in,
                           int cpad_out, int dtype, void* stream);
/* the same with the dgrad pack written into tap slots tap0 .. tap0+taps-1 of rows that hold row_taps taps:
 * dgrad [Ci][row_taps][cpad_out] (a 3x3 weight and its block's 1x1 downsample weight share one buffer, see A2 above) */
int htrvt_pack_conv_weight_slots(const float* w, void* fwd, void* dgrad, int Co, int Ci, int taps, int cpad_in, int cpad_out,
                                 int row_taps, int tap0, int dtype, void* stream);
/* grad [Co][Ci][taps] += packed [taps][cpad_in][Co]  (the conv-wgrad GEMM output) */
int htrvt_unpack_conv_wgrad(const float* packed, float* grad, int Co, int Ci, int taps, int cpad_in, void* stream);
int htrvt_cast_f32(const float* src, void* dst, int64_t n, int dtype, void* stream);

/* The same re-layouts as ONE launch over a table of jobs (csrc/relayout.hip): every conv / Linear weight of the model is
 * re-packed once per optimizer step and every conv weight gradient unpacked once per backward -- 47 small launches per step
 * as single-tensor calls.  Fill src / dst / kind / sizes, let htrvt_relayout_plan assign the workgroup ranges (host side),
 * copy the table to the device, then htrvt_relayout(table_dev, njobs, total) runs all jobs.  A job's fields mean what the
 * single-tensor entry point of its kind documents above / below. */
#define HTRVT_RELAYOUT_PACK_CONV 0      /* src w [d0 = Co][d1 = Ci][taps] f32 -> dst0 fwd pack, dst1 dgrad pack (either may be NULL) */
#define HTRVT_RELAYOUT_CAST_TRANSPOSE 1 /* src w [d0 = rows][d1 = cols] f32 -> dst0 [rows][cols] (may be NULL), dst1 [cols][cpad_in = ld_t] */
#define HTRVT_RELAYOUT_UNPACK_WGRAD 2   /* src packed [taps][cpad_in][d0 = Co] f32, dst0 grad [Co][d1 = Ci][taps] f32 += */
#define HTRVT_RELAYOUT_MAX_JOBS 64
typedef struct HtrvtRelayoutJob {
  const void* src;
  void* dst0;
  void* dst1;
  int kind;
  int d0, d1;
  int taps, cpad_in, cpad_out, row_taps, tap0;
  int tile0, tiles_x;     /* filled by htrvt_relayout_plan */
  int reserved_[2];
} HtrvtRelayoutJob;
int htrvt_relayout_plan(HtrvtRelayoutJob* jobs_host, int njobs);   /* returns the launch's workgroup count, < 0 on a bad job */
int htrvt_relayout(const HtrvtRelayoutJob* jobs_dev, int njobs, int total_tiles, int dtype, void* stream);
/* the same launch from a planned table in HOST memory, copied into the kernel-argument segment (no device copy of the
 * table to keep alive or to refresh when a gradient buffer moves); njobs <= HTRVT_RELAYOUT_ARG_JOBS */
#define HTRVT_RELAYOUT_ARG_JOBS 48
int htrvt_relayout_host(const HtrvtRelayoutJob* jobs_host, int njobs, int total_tiles, int dtype, void* stream);
/* Linear weight w [rows = out][cols = in] float32 -> dst [rows][cols] (may be NULL) and dst_t [cols][ld_t] = w^T in
 * `dtype` (bfloat16), columns rows .. ld_t-1 of dst_t zero: the K-major B operand of the Linear dgrad GEMM
 * dx[M][in] = dy[M][out] * w (HTR_VT.py:22-37 backward), so that forward and dgrad run the same kernel. */
int htrvt_cast_transpose_f32(const float* src, void* dst, void* dst_t, int rows, int cols, int ld_t, int dtype, void* stream);

/* ---- split-bfloat16 operands of the parity path (csrc/split.hip) -----------------------------------------------------
 * x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi); x*w = x_hi w_hi + x_lo w_hi + x_hi w_lo + O(2^-16 |x w|)
 * is ONE bfloat16 product over a three times longer K when the operands are concatenated along K (float32 accumulate):
 * the reference's float32 Linear / conv arithmetic (HTR_VT.py:22-37,76; resnet18.py:26-31,59-63) on the bf16 matrix
 * cores at a third of their rate instead of 1/16 (the f32-input MFMA), inside BASELINE.json's 1e-3 logit gate.
 * src float32 [rows][cols] (row stride ld_src) ->
 *   cat (may be NULL unless cat_f32 / transpose): [rows][3*cols], order 0 = (hi | lo | hi) -- the activation side,
 *       order 1 = (hi | hi | lo) -- the weight side; bfloat16, or float32 holding the same bf16-exact values when cat_f32
 *       (input of htrvt_pack_conv_weight); transpose: [cols][3*rows], block j of row c = part j of src[:, c]
 *   hi, lo (may be NULL): bfloat16 [rows][cols] planes (weight gradients contract over rows: three accumulating launches) */
int htrvt_split_bf16(const float* src, int64_t rows, int cols, int64_t ld_src, void* cat, int order, int cat_f32, int transpose,
                     void* hi, void* lo, void* stream);
/* Split-bf16 strided conv dgrad (resnet18.py:26,59-63 backward): the parity-class launches of htrvt_gemm with float32 output
 * leave one dense [B][Hq][Wq][C] matrix per class (a, b) = (hi % sh, wi % sw); this interleaves them into the NHWC gradient
 * dx [B][Hi][Wi][C] (+ residual, may be NULL).  c01 / c10 / c11 may be NULL where the stride in that direction is 1. */
/* float32 element-wise steps beside the split-bf16 Linear products (which write plain float32 + bias): mode 0: out = gelu_erf(a)
 * (timm Mlp's exact-erf GELU, HTR_VT.py:76), 1: out = a * gelu_erf'(b) (its backward), 2: out = a + b (residual, HTR_VT.py:81-82).
 * The float32 path has the same steps, in the same order and with the same rounding points, inside its GEMM epilogue. */
int htrvt_elementwise_f32(const float* a, const float* b, float* out, int64_t n, int mode, void* stream);
int htrvt_class_scatter_f32(const float* c00, const float* c01, const float* c10, const float* c11, const float* residual, float* dx,
                            int B, int Hi, int Wi, int C, int sh, int sw, void* stream);

/* ---- optimizer step (train.py:94 AdamW(betas .9/.99, wd .5) as one flat launch) -- */
/* Statement order and rounding points of torch.optim.AdamW's single-tensor step; the hyper-parameters are doubles as in
 * Python, derived scalars (1 - lr*wd, 1 - beta, bias corrections, -lr/bc1) are formed in double and rounded once. */
int htrvt_adamw(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                double eps, double weight_decay, int step, void* stream);
/* The same step with its derived scalars read from DEVICE memory: a captured HIP graph bakes kernel arguments, while lr
 * (utils.update_lr_cos, train.py:115) and the step number change every iteration.  htrvt_adamw_scalars forms the
 * HTRVT_ADAMW_SCALARS floats on the host exactly as htrvt_adamw does (same doubles, same single rounding), the caller
 * copies them to `hyper_dev` in stream order before the launch / the graph replay: bit-identical to htrvt_adamw. */
#define HTRVT_ADAMW_SCALARS 8
int htrvt_adamw_scalars(double lr, double beta1, double beta2, double eps, double weight_decay, int step, float* out_host);
int htrvt_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, void* stream);

/* ---- around the path: SAM, EMA, greedy decode ---------------- */
/* out[0] = sum x^2 (deterministic two-stage); partial: htrvt_sumsq_blocks(n) floats of scratch.  With x = the flat
 * gradient buffer this is SAM._grad_norm()^2 (utils/sam.py:47-56). */
int htrvt_sumsq_blocks(int64_t n);
int htrvt_sumsq(const float* x, int64_t n, float* partial, float* out, void* stream);
/* SAM.first_step (utils/sam.py:15-27, adaptive=False): old_p = p; p += g * rho / (sqrt(norm_sq[0]) + 1e-12) */
int htrvt_sam_first_step(float* p, const float* g, float* old_p, int64_t n, float rho, const float* norm_sq, void* stream);
/* SAM.second_step's restore (utils/sam.py:31-34): p = old_p; htrvt_adamw then applies the base optimizer */
int htrvt_sam_restore(float* p, const float* old_p, int64_t n, void* stream);
/* ModelEma.update (utils/utils.py:158-173) over every state_dict entry in one launch: ema = ema*decay + (1-decay)*model
 * in float32 (int64 entries: float math, truncated on the way back).  table: device array of `count` entries. */
typedef struct HtrvtEmaEntry {
  void* ema;
  const void* model;
  int64_t numel;
  int32_t is_int64;
  int32_t pad_;
} HtrvtEmaEntry;
int htrvt_ema_update(const HtrvtEmaEntry* table, int count, int64_t max_numel, double decay, void* stream);
/* valid.py:40-42 + CTCLabelConverter.decode (utils/utils.py:72-86): out[b, 0:out_len[b]] = arg-max class per frame with
 * blanks (0), repeats and indices >= ncharacter removed.  logits [B,T,ld>=C] float32 (arg-max of the logits = arg-max of
 * their log-softmax), out [B,T] int32. */
int htrvt_ctc_greedy_decode(const float* logits, int B, int T, int C, int64_t ld, int ncharacter, int32_t* out,
                            int32_t* out_len, void* stream);

/* ---- in front of the path: the loader's per-image preparation -------------------------------------
 * data/dataset.py:104-135 for a ragged batch of grey uint8 scans: aspect-preserving resize to height H (npThum: width' =
 * min(int(w * H / h), W), PIL.Image.resize = Pillow's 8-bit BICUBIC resampler, restated bit for bit), img_as_float32 and
 * right pad with 1.0 -- delivered as the uint8 batch dst [B][H][W] (255 = 1.0) that the model reads as value / 255.
 * src: all scans back to back; table[i] (device) = {byte offset of scan i in src, byte offset of its scratch rows in tmp
 * (h_i * W bytes each), h_i, w_i}; max_src_h = max h_i.  Every scale factor h_i / H and w_i / width' must be
 * <= htrvt_line_max_scale() (the tap table of one output pixel is bounded); an image beyond it (or with h, w or width'
 * < 1) is delivered as an empty line, all 255, never read past the table. */
typedef struct HtrvtLineImage {
  int64_t src_offset;
  int64_t tmp_offset;
  int32_t h, w;
} HtrvtLineImage;
int htrvt_line_max_scale(void);
int htrvt_line_prepare(const uint8_t* src, const HtrvtLineImage* table, uint8_t* tmp, uint8_t* dst, int B, int H, int W,
                       int max_src_h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HTRVT_H */
