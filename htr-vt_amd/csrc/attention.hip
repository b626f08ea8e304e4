// attention.hip -- fused multi-head self-attention for the HTR-VT encoder block (bfloat16 throughput path, gfx950).
//
// Replaces, per transformer block and direction, the six launches and two [B*h,N,N] HBM round trips of
//   attn = softmax(q @ k^T * scale); x = attn @ v                 (reference model_v1/model/HTR_VT.py:27-36, scale :17)
// and of their autograd backward (train.py:123) by ONE launch each.  The scores S and probabilities P never leave the
// CU: S tiles live in MFMA accumulators, P is rounded to bfloat16 in registers and fed straight back as an MFMA
// operand (accumulator-as-operand k order), K / V (forward) and Q / dO (backward) tiles are staged through LDS.
//
// Layouts (what the qkv Linear writes, HTR_VT.py:29-30): qkv [B*N][3][h][hd] bfloat16, out / dout [B*N][h][hd].
// lse2 [B][h][N] float32 = log2 of the softmax denominator in the scaled base-2 domain:
//   P[q][k] = exp2(S[q][k] * scale * log2(e) - lse2[q]),   saved by the forward for the recomputing backward.
//
// CDNA4 mapping (forward; the backward kernels follow the same scheme with the roles of rows / lanes swapped):
//   * workgroup = 4 waves = 128 queries of one (batch, head); two workgroups per CU (64 KB LDS, <= 256 VGPRs each).
//   * "swapped" products: S^T = K Q^T and O^T = V^T P^T with v_mfma_f32_32x32x16_bf16, so a lane always owns ONE query
//     (column of the accumulator tile): row max / row sum are 31 in-register ops + one cross-half exchange, the online
//     softmax rescale is a per-lane scalar, and P^T (keys in the accumulator registers) is the B operand of the second
//     product without any lane movement.
//   * K is read by rows (ds_read_b128), V transposed (ds_read_b64_tr_b16); both tiles use one XOR-swizzled image that
//     is conflict-free for both kinds of read (tools/lds_bank_check.py applies the banking rules to it).
//   * K/V tiles (64 keys) are double buffered: global loads for tile t+1 are issued before the MFMAs of tile t and
//     written to LDS after them (one barrier per tile).
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"

using namespace htrvt;

namespace {

constexpr int KT = 64;          // keys (forward, dQ role) or queries (dK/dV role) per staged tile
constexpr float LOG2E = 1.44269504088896340736f;

struct AttnParams {
  const bf16_t* qkv;
  bf16_t* out;         // forward output [B*N][h*hd]
  float* lse2;         // [B*h][N]
  const bf16_t* dout;  // backward: gradient of out
  const bf16_t* o;     // backward: forward output
  bf16_t* dqkv;        // backward: gradient of qkv
  const float* bias;   // [h][N][N] additive score bias (natural-log units, added after the scale) or NULL
  float* dbias;        // backward: [h][N][N] += sum_b dS (float atomics) or NULL
  int B, N, h;
  float sl2;           // scale * log2(e)
  float scale;
};

// byte offset of 16-byte chunk `ch` of row `row` in a [rows][HD] bfloat16 LDS tile
template <int HD>
__device__ __forceinline__ int lds_off(int row, int ch) {
  if constexpr (HD == 128) return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
  else if constexpr (HD == 64) return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)));
  else return 64 * row + 16 * (ch ^ ((row >> 2) & 3));
}

typedef __attribute__((address_space(3))) s16x4_t* lds_tr_ptr;

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1.  Element indices of the accumulator vectors are
// constants from the start this way; with `#pragma unroll` loops whose body holds a store or an atomic the vectors were
// indexed dynamically for a while and ended up in scratch memory.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Per-lane address constants of the two fragment reads.  lds_off's XOR term depends only on the low four bits of the tile
// row, which are lane bits in both reads (row0 is a multiple of 32, 16 s a multiple of 16), and the chunk index is a
// compile-time part OR-ed with a lane part on disjoint bits -- so every fragment address is
//     (lane constant) XOR (compile-time constant) + (compile-time constant),
// one v_xor per distinct (s) / (dt) instead of the full shift / mask / xor chain per read (the kernels are VALU-bound
// beside their MFMAs).  tools/lds_bank_check.py checks these forms against lds_off for every lane.
template <int HD>
struct LaneAddr {
  int rowb;        // row read:  lds_off(row0 + r, 2 s + h)              = (rowb ^ 32 s) + ROWB row0
  int trb0, trb1;  // transposed: lds_off(row0 + 16 s + 4 h + q [+ 8], 4 dt + 2 g + (p >> 1)) + 8 (p & 1)
                   //                                                    = (trb ^ 64 dt) + ROWB (row0 + 16 s)
};

template <int HD>
__device__ __forceinline__ LaneAddr<HD> lane_addr(int lane) {
  const int h = lane >> 5, g = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
  LaneAddr<HD> la;
  la.rowb = lds_off<HD>(lane & 31, h);
  la.trb0 = lds_off<HD>(4 * h + q, 2 * g + (p >> 1)) + 8 * (p & 1);
  la.trb1 = lds_off<HD>(4 * h + q + 8, 2 * g + (p >> 1)) + 8 * (p & 1);
  return la;
}

// A operand (rows = 32 consecutive columns of the tile starting at 32*dt, k = 16 tile rows in accumulator-as-operand
// order: element j of lane half h is tile row row0 + 16 s + 8 (j >> 2) + 4 h + (j & 3)) by two transposed reads
template <int HD>
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, const LaneAddr<HD>& la, int row0, int s, int dt) {
  constexpr int ROWB = HD * 2;
  const int add = ROWB * (row0 + 16 * s);
  const s16x4_t r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(tile + ((la.trb0 ^ (64 * dt)) + add)));
  const s16x4_t r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(tile + ((la.trb1 ^ (64 * dt)) + add)));
  const s16x8_t r = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  return __builtin_bit_cast(bf16x8_t, r);
}

// row operand: lane (r = lane & 31, h = lane >> 5) takes elements 16 s + 8 h .. + 7 of tile row row0 + r
template <int HD>
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, const LaneAddr<HD>& la, int row0, int s) {
  constexpr int ROWB = HD * 2;
  const uint4 v = *reinterpret_cast<const uint4*>(tile + ((la.rowb ^ (32 * s)) + ROWB * row0));
  return __builtin_bit_cast(bf16x8_t, v);
}

// registers 8 s .. 8 s + 7 of a 32x32 accumulator tile, rounded to bfloat16: the operand of a following MFMA that
// contracts over the tile's ROW index
__device__ __forceinline__ bf16x8_t acc_frag(const f32x16_t& x, int s) {
  uint4 v;
  v.x = pack_bf16x2(x[8 * s + 0], x[8 * s + 1]);
  v.y = pack_bf16x2(x[8 * s + 2], x[8 * s + 3]);
  v.z = pack_bf16x2(x[8 * s + 4], x[8 * s + 5]);
  v.w = pack_bf16x2(x[8 * s + 6], x[8 * s + 7]);
  return __builtin_bit_cast(bf16x8_t, v);
}

// staging of one [KT][HD] tile: global -> registers (issue) ... registers -> LDS (commit), NTH threads
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));   // a first-class 16-byte vector: HIP's uint4 is a struct whose
                                                                    // copies are memcpys, which kept the staged tile in scratch
                                                                    // memory whenever a store or an atomic sat in between
template <int HD, int NTH, int ROWS = KT>
struct TileStage {
  static constexpr int CPR = HD / 8;                 // 16-byte chunks per row
  static constexpr int NL = ROWS * CPR / NTH;        // loads per thread
  static_assert(ROWS * CPR % NTH == 0 && NL >= 1 && NL <= 8, "tile must divide over the threads, at most 8 loads each");
  u32x4_t r0, r1, r2, r3, r4, r5, r6, r7;

  static __device__ __forceinline__ int row_of(int u) { return (threadIdx.x + NTH * u) / CPR; }
  static __device__ __forceinline__ int ch_of(int u) { return (threadIdx.x + NTH * u) % CPR; }
  static __device__ __forceinline__ u32x4_t ld16(const bf16_t* base, long long ld, int row, int ch) {
    return *reinterpret_cast<const u32x4_t*>(base + (long long)row * ld + ch * 8);
  }
  static __device__ __forceinline__ void st16(char* tile, int u, const u32x4_t& v) {
    *reinterpret_cast<u32x4_t*>(tile + lds_off<HD>(row_of(u), ch_of(u))) = v;
  }

  // rows past `last` (the sequence's last token: a partial final tile) re-read that row -- always valid memory; whatever
  // they contribute is masked by the caller (scores of padding keys -> -inf / probabilities of padding queries -> 0)
  __device__ __forceinline__ void issue(const bf16_t* base, long long ld, int row0, int last) {
    r0 = ld16(base, ld, min(row0 + row_of(0), last), ch_of(0));
    if constexpr (NL > 1) r1 = ld16(base, ld, min(row0 + row_of(1), last), ch_of(1));
    if constexpr (NL > 2) r2 = ld16(base, ld, min(row0 + row_of(2), last), ch_of(2));
    if constexpr (NL > 3) r3 = ld16(base, ld, min(row0 + row_of(3), last), ch_of(3));
    if constexpr (NL > 4) r4 = ld16(base, ld, min(row0 + row_of(4), last), ch_of(4));
    if constexpr (NL > 5) r5 = ld16(base, ld, min(row0 + row_of(5), last), ch_of(5));
    if constexpr (NL > 6) r6 = ld16(base, ld, min(row0 + row_of(6), last), ch_of(6));
    if constexpr (NL > 7) r7 = ld16(base, ld, min(row0 + row_of(7), last), ch_of(7));
  }
  __device__ __forceinline__ void commit(char* tile) const {
    st16(tile, 0, r0);
    if constexpr (NL > 1) st16(tile, 1, r1);
    if constexpr (NL > 2) st16(tile, 2, r2);
    if constexpr (NL > 3) st16(tile, 3, r3);
    if constexpr (NL > 4) st16(tile, 4, r4);
    if constexpr (NL > 5) st16(tile, 5, r5);
    if constexpr (NL > 6) st16(tile, 6, r6);
    if constexpr (NL > 7) st16(tile, 7, r7);
  }
};

// v_exp_f32 directly: every argument here is <= ~0 (a score minus its row maximum / log-sum-exp), results below 2^-126
// flush to zero, which is what a probability that small is worth; exp2f() would wrap the instruction in range scaling
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }   // the other 32-lane half's value

// O[row][d] of a lane-per-row accumulator set: lane (r, hf) owns memory row `rowptr`, accumulator tile d register i is
// column 32 d + (i & 3) + 8 (i >> 2) + 4 hf: four consecutive columns per register quad -> 8-byte stores
template <int ND>
__device__ __forceinline__ void store_lane_rows(const f32x16_t (&acc)[ND], bf16_t* rowptr, int hf, float mul) {
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 v;
      v.x = pack_bf16x2(acc[d][4 * g + 0] * mul, acc[d][4 * g + 1] * mul);
      v.y = pack_bf16x2(acc[d][4 * g + 2] * mul, acc[d][4 * g + 3] * mul);
      *reinterpret_cast<uint2*>(rowptr + 32 * d + 8 * g + 4 * hf) = v;
    }
}

// -------------------------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------------------------
// DBG: compile-time timing ablations (results are wrong when != 0; instantiate by hand for an experiment): 1 no global
// loads inside the key loop, 2 no softmax arithmetic, 4 no P V product.  Measured at B=128, N=256, hd=128:
// 58.8 us as shipped, 50.6 without the softmax, 36.4 without softmax and P V; the HBM floor of the launch is 32 us.
template <int HD, int DBG = 0, bool BIAS = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
  constexpr int NTH = 256, QB = 128;
  constexpr int TILE_B = KT * HD * 2;
  constexpr int NS = HD / 16;       // k-steps of the QK^T product
  constexpr int ND = HD / 32;       // 32-row tiles of O^T
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]

  const int nqb = (p.N + QB - 1) / QB;
  const int last = p.N - 1;
  const int total = gridDim.x;
  int id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);   // the query blocks of one head share an XCD (K/V in its L2)
  const int bh = id / nqb, qb = id - bh * nqb;
  const int b = bh / p.h, hh = bh - b * p.h;
  const long long ld = 3ll * p.h * HD;
  const bf16_t* qbase = p.qkv + (long long)b * p.N * ld + hh * HD;
  const bf16_t* kbase = qbase + p.h * HD;
  const bf16_t* vbase = kbase + p.h * HD;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const LaneAddr<HD> la = lane_addr<HD>(lane);
  const int q0 = qb * QB + wave * 32;
  const int qrow = min(q0 + r, last);      // a padding query of the last block re-reads the last token; its row is not stored

  // Q^T as the B operand of S^T = K Q^T: lane (r, hf) holds Q[q0 + r][16 s + 8 hf .. + 7]
  bf16x8_t qf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s)
    qf[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qbase + (long long)qrow * ld + 16 * s + 8 * hf));

  TileStage<HD, NTH> sk, sv;
  sk.issue(kbase, ld, 0, last);
  sv.issue(vbase, ld, 0, last);
  sk.commit(smem);
  sv.commit(smem + TILE_B);
  __syncthreads();

  f32x16_t o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
  float m = -INFINITY, l = 0.f;     // running max (scaled base-2 domain) and this lane half's share of the running sum

  const int nt = (p.N + KT - 1) / KT;
  const bool ragged = (p.N % KT) != 0;      // the last key tile holds padding keys
  for (int t = 0; t < nt; ++t) {
    const char* kt = smem + (t & 1) * 2 * TILE_B;
    const char* vt = kt + TILE_B;
    char* nxt = smem + ((t + 1) & 1) * 2 * TILE_B;
    // the last iteration re-stages its own tile into the idle buffer (nothing reads it): no conditional around the
    // loads, so the staging registers stay registers
    const int tn = min(t + 1, nt - 1);
    if constexpr (!(DBG & 1)) {
      sk.issue(kbase, ld, tn * KT, last);
      sv.issue(vbase, ld, tn * KT, last);
    }
    // S^T tiles: keys 32 c .. 32 c + 31 of this tile x the wave's 32 queries
    f32x16_t st[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int i = 0; i < 16; ++i) st[c][i] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s)
        st[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(kt, la, 32 * c, s), qf[s], st[c], 0, 0, 0);
    }
    if (!BIAS && ragged && t == nt - 1) {   // workgroup-uniform: scores of the padding keys -> -inf (probability 0)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (t * KT + 32 * c + (i & 3) + 8 * (i >> 2) + 4 * hf > last) st[c][i] = -INFINITY;
    }
    // online softmax for query r: this lane holds 32 of the tile's 64 keys, lane ^ 32 the other 32
    if constexpr (!(DBG & 2)) {
    float mx = -INFINITY;
    if constexpr (BIAS) {
      // score = S * scale + bias[h][q][k] (relative-position bias, window / padding mask as a large negative number):
      // this lane's query row, four consecutive keys per accumulator register quad
      const float* brow = p.bias + ((long long)hh * p.N + q0 + r) * p.N + t * KT + 4 * hf;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(brow + 32 * c + 8 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            st[c][4 * g + j] = fmaf(st[c][4 * g + j], p.sl2, (&bv.x)[j] * LOG2E);
            mx = fmaxf(mx, st[c][4 * g + j]);
          }
        }
      mx = fmaxf(mx, xhalf(mx));
    } else {                  // maximum of the raw scores (sl2 > 0: scaling commutes with the maximum)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[c][i]);
      mx = fmaxf(mx, xhalf(mx)) * p.sl2;
    }
    const float mn = fmaxf(m, mx);
    float rs = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        st[c][i] = BIAS ? fast_exp2(st[c][i] - mn) : fast_exp2(fmaf(st[c][i], p.sl2, -mn));
        rs += st[c][i];
      }
    if (__any(mn > m)) {      // wave-uniform: the running maximum of some query moved -> rescale what is accumulated
      const float alpha = fast_exp2(m - mn);
      m = mn;
      l *= alpha;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
    }
    l += rs;
    }
    // O^T += V^T P^T
    if constexpr (!(DBG & 4))
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8_t pb = acc_frag(st[c], s);
#pragma unroll
        for (int d = 0; d < ND; ++d)
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(vt, la, 32 * c, s, d), pb, o[d], 0, 0, 0);
      }
    sk.commit(nxt);
    sv.commit(nxt + TILE_B);
    __syncthreads();
  }

  l += xhalf(l);
  const float inv = 1.0f / l;
  if (q0 + r <= last) {
    store_lane_rows<ND>(o, p.out + ((long long)b * p.N + q0 + r) * ((long long)p.h * HD) + hh * HD, hf, inv);
    if (hf == 0 && p.lse2 != nullptr) p.lse2[(long long)bh * p.N + q0 + r] = m + log2f(l);
  }
}

// -------------------------------------------------------------------------------------------------------------------
// backward, first launch: dQ (and delta = rowsum(dO * O) for the second launch).  Same orientation as the forward: a
// lane owns one query, K / V tiles stream through LDS, P is recomputed from the saved lse2.
//   S^T = K Q^T ; P^T = exp2(S^T sl2 - lse2[q]) ; dP^T = V dO^T ; dS^T = P^T (dP^T - delta[q]) scale ; dQ^T += K^T dS^T
// -------------------------------------------------------------------------------------------------------------------
template <int HD, bool BIAS = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnParams p, float* __restrict__ delta) {
  constexpr int NTH = 256, QB = 128;
  constexpr int TILE_B = KT * HD * 2;
  constexpr int NS = HD / 16, ND = HD / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]

  const int nqb = (p.N + QB - 1) / QB;
  const int last = p.N - 1;
  const int total = gridDim.x;
  int id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int bh = id / nqb, qb = id - bh * nqb;
  const int b = bh / p.h, hh = bh - b * p.h;
  const long long ld = 3ll * p.h * HD, ldo = (long long)p.h * HD;
  const bf16_t* qbase = p.qkv + (long long)b * p.N * ld + hh * HD;
  const bf16_t* kbase = qbase + p.h * HD;
  const bf16_t* vbase = kbase + p.h * HD;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const LaneAddr<HD> la = lane_addr<HD>(lane);
  const int q0 = qb * QB + wave * 32;

  bf16x8_t qf[NS], dof[NS];
  float dl = 0.f;
  const int qrow = min(q0 + r, last);
  {
    const bf16_t* dorow = p.dout + ((long long)b * p.N + qrow) * ldo + hh * HD;
    const bf16_t* orow = p.o + ((long long)b * p.N + qrow) * ldo + hh * HD;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      qf[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qbase + (long long)qrow * ld + 16 * s + 8 * hf));
      Vec16<bf16_t> vd, vo;
      vd.raw = *reinterpret_cast<const uint4*>(dorow + 16 * s + 8 * hf);
      vo.raw = *reinterpret_cast<const uint4*>(orow + 16 * s + 8 * hf);
      dof[s] = __builtin_bit_cast(bf16x8_t, vd.raw);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl = fmaf(vd.get(j), vo.get(j), dl);
    }
  }
  dl += xhalf(dl);                                    // delta[q] = sum_d dO[q][d] O[q][d]
  const float lse = p.lse2[(long long)bh * p.N + qrow];
  if (hf == 0 && q0 + r <= last) delta[(long long)bh * p.N + q0 + r] = dl;

  TileStage<HD, NTH> sk, sv;
  sk.issue(kbase, ld, 0, last);
  sv.issue(vbase, ld, 0, last);
  sk.commit(smem);
  sv.commit(smem + TILE_B);
  __syncthreads();

  f32x16_t dq[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;

  const int nt = (p.N + KT - 1) / KT;
  const bool ragged = (p.N % KT) != 0;
  for (int t = 0; t < nt; ++t) {
    const char* kt = smem + (t & 1) * 2 * TILE_B;
    const char* vt = kt + TILE_B;
    char* nxt = smem + ((t + 1) & 1) * 2 * TILE_B;
    const int tn = min(t + 1, nt - 1);
    sk.issue(kbase, ld, tn * KT, last);
    sv.issue(vbase, ld, tn * KT, last);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      f32x16_t st, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) st[i] = dp[i] = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(kt, la, 32 * c, s), qf[s], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(vt, la, 32 * c, s), dof[s], dp, 0, 0, 0);
      }
      if constexpr (BIAS) {
        const float* brow = p.bias + ((long long)hh * p.N + q0 + r) * p.N + t * KT + 32 * c + 4 * hf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(brow + 8 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float pr = fast_exp2(fmaf(st[4 * g + j], p.sl2, fmaf((&bv.x)[j], LOG2E, -lse)));
            st[4 * g + j] = pr * (dp[4 * g + j] - dl) * p.scale;
          }
        }
      } else {
        if (ragged && t == nt - 1) {       // workgroup-uniform: scores of the padding keys -> -inf, their probability is 0
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (t * KT + 32 * c + (i & 3) + 8 * (i >> 2) + 4 * hf > last) st[i] = -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float pr = fast_exp2(fmaf(st[i], p.sl2, -lse));
          st[i] = pr * (dp[i] - dl) * p.scale;          // dS^T
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8_t dsb = acc_frag(st, s);
#pragma unroll
        for (int d = 0; d < ND; ++d)
          dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(kt, la, 32 * c, s, d), dsb, dq[d], 0, 0, 0);
      }
    }
    sk.commit(nxt);
    sv.commit(nxt + TILE_B);
    __syncthreads();
  }
  if (q0 + r <= last) store_lane_rows<ND>(dq, p.dqkv + ((long long)b * p.N + q0 + r) * ld + hh * HD, hf, 1.0f);
}

// -------------------------------------------------------------------------------------------------------------------
// backward, second launch: dK and dV.  A lane owns one KEY (K / V fragments of the wave's 32 keys stay in registers),
// Q / dO tiles stream through LDS (read by rows for S and dP, transposed for dV^T and dK^T), queries sit in the
// accumulator rows, so the per-query constants lse2 / delta are per-register values read (broadcast) from LDS.
//   S = Q K^T ; P = exp2(S sl2 - lse2[q]) ; dP = dO V^T ; dS = P (dP - delta[q]) scale ; dV^T += dO^T P ; dK^T += Q^T dS
// One wave per SIMD (the two 32 x HD accumulator sets + the K / V fragments need > 256 registers).
// -------------------------------------------------------------------------------------------------------------------
template <int HD, bool BIAS = false>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv_kernel(const AttnParams p, const float* __restrict__ delta) {
  // 128-query tiles here (the forward / dQ kernels stage 64 keys): one workgroup per CU leaves 160 KB of LDS, and at
  // N = 256 the whole pass is two tiles -- the second one in flight under the first one's 128 MFMAs per wave
  constexpr int NTH = 256, KB = 128, QT = HD >= 128 ? 64 : 128;   // measured: 128-query tiles pay at hd 64 (286 vs 328 us), not at hd 128 (254 vs 244)
  constexpr int TILE_B = QT * HD * 2;
  constexpr int STAGE_B = 2 * TILE_B + 2 * QT * 4 + 16;     // Q tile | dO tile | lse2[QT] | delta[QT] | dump word
  constexpr int NS = HD / 16, ND = HD / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nkb = (p.N + KB - 1) / KB;
  const int last = p.N - 1;
  const int total = gridDim.x;
  int id = blockIdx.x;
  if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
  const int bh = id / nkb, kb = id - bh * nkb;
  const int b = bh / p.h, hh = bh - b * p.h;
  const long long ld = 3ll * p.h * HD, ldo = (long long)p.h * HD;
  const bf16_t* qbase = p.qkv + (long long)b * p.N * ld + hh * HD;
  const bf16_t* kbase = qbase + p.h * HD;
  const bf16_t* vbase = kbase + p.h * HD;
  const bf16_t* dobase = p.dout + (long long)b * p.N * ldo + hh * HD;
  const float* lsebase = p.lse2 + (long long)bh * p.N;
  const float* delbase = delta + (long long)bh * p.N;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const LaneAddr<HD> la = lane_addr<HD>(lane);
  const int k0 = kb * KB + wave * 32;

  bf16x8_t kf[NS], vf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    kf[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(kbase + (long long)min(k0 + r, last) * ld + 16 * s + 8 * hf));
    vf[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(vbase + (long long)min(k0 + r, last) * ld + 16 * s + 8 * hf));
  }

  TileStage<HD, NTH, QT> sq, sd;
  // thread < 64: lse2 of query tid of the staged tile; 64 <= thread < 128: delta of query tid - 64 (other threads
  // re-read entry 0: no branch around the load)
  const float* cbase = threadIdx.x < QT ? lsebase + threadIdx.x : (threadIdx.x < 2 * QT ? delbase + (threadIdx.x - QT) : lsebase);
  const int cslot = threadIdx.x < 2 * QT ? threadIdx.x : 2 * QT;       // slot 2*QT: a dump word behind the two arrays
  const int cidx = threadIdx.x < QT ? (int)threadIdx.x : (threadIdx.x < 2 * QT ? (int)threadIdx.x - QT : 0);   // query of the tile this thread's constant belongs to
  float sc;
  sq.issue(qbase, ld, 0, last);
  sd.issue(dobase, ldo, 0, last);
  const bool is_lse = threadIdx.x < QT;
  sc = cbase[min(cidx, last) - cidx];
  if (is_lse && cidx > last) sc = INFINITY;       // padding query of a partial tile: exp2(s - inf) = 0, no NaN (s is finite)
  sq.commit(smem);
  sd.commit(smem + TILE_B);
  reinterpret_cast<float*>(smem + 2 * TILE_B)[cslot] = sc;
  __syncthreads();

  f32x16_t dk[ND], dv[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) dk[d][i] = dv[d][i] = 0.f;

  const int nt = (p.N + QT - 1) / QT;
  const bool ragged = (p.N % QT) != 0;      // the last query tile holds padding queries
  for (int t = 0; t < nt; ++t) {
    const char* qt = smem + (t & 1) * STAGE_B;
    const char* dot = qt + TILE_B;
    const float* cst = reinterpret_cast<const float*>(qt + 2 * TILE_B);
    char* nxt = smem + ((t + 1) & 1) * STAGE_B;
    const int tn = min(t + 1, nt - 1);
    sq.issue(qbase, ld, tn * QT, last);
    sd.issue(dobase, ldo, tn * QT, last);
    sc = cbase[min(tn * QT + cidx, last) - cidx];
    if (is_lse && tn * QT + cidx > last) sc = INFINITY;
#pragma unroll
    for (int c = 0; c < QT / 32; ++c) {
      f32x16_t st, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) st[i] = dp[i] = 0.f;
      // One wave per SIMD: nothing but this wave's own instruction stream covers an LDS read, so the fragments of a whole
      // product are requested first and the MFMAs follow (read -> wait -> MFMA one at a time cost ~150 cycles per MFMA)
      // the tile's per-query constants (lse2, delta) of this 32-query block: requested ahead of the MFMAs that hide them
      // (the bias variant has no registers to spare for either: it keeps the one-at-a-time order)
      float4 lsq[4], deq[4];
      if constexpr (!BIAS) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          lsq[g] = *reinterpret_cast<const float4*>(cst + 32 * c + 8 * g + 4 * hf);
          deq[g] = *reinterpret_cast<const float4*>(cst + QT + 32 * c + 8 * g + 4 * hf);
        }
      }
      constexpr int NSG = BIAS ? 1 : (NS >= 4 ? 4 : NS);      // k-steps whose fragments are requested together
#pragma unroll
      for (int s0 = 0; s0 < NS; s0 += NSG) {
        bf16x8_t fq[NSG], fd[NSG];
#pragma unroll
        for (int s = 0; s < NSG; ++s) {
          fq[s] = row_frag<HD>(qt, la, 32 * c, s0 + s);
          fd[s] = row_frag<HD>(dot, la, 32 * c, s0 + s);
        }
#pragma unroll
        for (int s = 0; s < NSG; ++s) {
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[s], kf[s0 + s], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[s], vf[s0 + s], dp, 0, 0, 0);
        }
      }
      // accumulator register i is query 32 c + (i & 3) + 8 (i >> 2) + 4 hf of the tile
      static_for<0, 4>([&](auto G) {
        constexpr int g = decltype(G)::value;
        const float4 ls = BIAS ? *reinterpret_cast<const float4*>(cst + 32 * c + 8 * g + 4 * hf) : lsq[g];
        const float4 de = BIAS ? *reinterpret_cast<const float4*>(cst + QT + 32 * c + 8 * g + 4 * hf) : deq[g];
        const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
        static_for<0, 4>([&](auto J) {
          constexpr int j = decltype(J)::value, i = 4 * g + j;
          float sb = -lsv[j];
          long long boff = 0;
          if constexpr (BIAS) {     // bias[h][query][key]: lanes of a half are 32 consecutive keys of one query row
            boff = ((long long)hh * p.N + t * QT + 32 * c + 8 * g + 4 * hf + j) * p.N + k0 + r;
            sb = fmaf(p.bias[boff], LOG2E, sb);
          }
          const float pr = fast_exp2(fmaf(st[i], p.sl2, sb));                // (a padding query carries lse2 = +inf: pr = 0)
          const float dsu = pr * (dp[i] - dev[j]);                          // d(score): gradient of the bias entry too
          if constexpr (BIAS) {   // summed over the batch (skipped when the bias is a constant mask: dbias == NULL)
            typedef __attribute__((address_space(1))) float gfloat;
            if (p.dbias != nullptr) __hip_atomic_fetch_add((gfloat*)(p.dbias + boff), dsu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          st[i] = pr;                                                       // P
          dp[i] = dsu * p.scale;                                            // dS
        });
      });
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8_t pb = acc_frag(st, s), dsb = acc_frag(dp, s);
        if constexpr (!BIAS) {
          bf16x8_t tv[ND], tk[ND];
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            tv[d] = tr_frag<HD>(dot, la, 32 * c, s, d);
            tk[d] = tr_frag<HD>(qt, la, 32 * c, s, d);
          }
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tv[d], pb, dv[d], 0, 0, 0);
            dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tk[d], dsb, dk[d], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(dot, la, 32 * c, s, d), pb, dv[d], 0, 0, 0);
            dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(qt, la, 32 * c, s, d), dsb, dk[d], 0, 0, 0);
          }
        }
      }
    }
    sq.commit(nxt);
    sd.commit(nxt + TILE_B);
    reinterpret_cast<float*>(nxt + 2 * TILE_B)[cslot] = sc;
    __syncthreads();
  }
  if (k0 + r <= last) {
    bf16_t* grow = p.dqkv + ((long long)b * p.N + k0 + r) * ld + hh * HD;
    store_lane_rows<ND>(dk, grow + p.h * HD, hf, 1.0f);
    store_lane_rows<ND>(dv, grow + 2 * p.h * HD, hf, 1.0f);
  }
}

template <typename K>
int set_lds(K kern, int smem, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(%d B LDS): %s", what, smem, hipGetErrorString(e));
    return -2;
  }
  return 0;
}

template <int HD, bool BIAS>
int launch_fwd(const AttnParams& p, hipStream_t st) {
  constexpr int smem = 2 * 2 * KT * HD * 2;
  static bool attr_done = false;
  auto kern = attn_fwd_kernel<HD, 0, BIAS>;
  if (!attr_done) {
    if (int rc = set_lds(kern, smem, "attn_fwd")) return rc;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.B * p.h * ((p.N + 127) / 128)), dim3(256), smem, st, p);
  return check_launch("attn_fwd");
}

template <int HD, bool BIAS>
int launch_bwd(const AttnParams& p, float* delta, hipStream_t st) {
  constexpr int smem_dq = 2 * 2 * KT * HD * 2;
  constexpr int QT = HD >= 128 ? 64 : 128;
  constexpr int smem_kv = 2 * (2 * QT * HD * 2 + 2 * QT * 4 + 16);
  static bool attr_done = false;
  auto kq = attn_bwd_dq_kernel<HD, BIAS>;
  auto kkv = attn_bwd_dkv_kernel<HD, BIAS>;
  if (!attr_done) {
    if (int rc = set_lds(kq, smem_dq, "attn_bwd_dq")) return rc;
    if (int rc = set_lds(kkv, smem_kv, "attn_bwd_dkv")) return rc;
    attr_done = true;
  }
  const dim3 grid(p.B * p.h * ((p.N + 127) / 128));
  hipLaunchKernelGGL(kq, grid, dim3(256), smem_dq, st, p, delta);
  hipLaunchKernelGGL(kkv, grid, dim3(256), smem_kv, st, p, (const float*)delta);
  return check_launch("attn_bwd");
}

}  // namespace

// any sequence length >= 32 (a partial last tile is masked in the kernels); with a score bias: multiples of 128 only
// (the bias rows are read in whole tiles)
extern "C" int htrvt_attn_supported(int N, int hd, int dtype) {
  return dtype == HTRVT_BF16 && N >= 32 && (hd == 32 || hd == 64 || hd == 128);
}

extern "C" int htrvt_attn_fwd(const void* qkv, const float* bias, void* out, float* lse2, int B, int N, int heads, int hd,
                              float scale, int dtype, void* stream) {
  HTRVT_REQUIRE(qkv && out, "htrvt_attn_fwd: null operand");
  HTRVT_REQUIRE(B > 0 && heads > 0 && htrvt_attn_supported(N, hd, dtype),
                "htrvt_attn_fwd: unsupported shape/dtype (N=%d >= 32, hd=%d in {32,64,128}, bfloat16)", N, hd);
  HTRVT_REQUIRE(bias == nullptr || N % 128 == 0, "htrvt_attn_fwd: with a score bias N=%d must be a multiple of 128 (pad, masking the padding keys in the bias)", N);
  HTRVT_REQUIRE((long long)B * N * 3 * heads * hd < (1ll << 31), "htrvt_attn_fwd: qkv too large");
  AttnParams p{};
  p.qkv = (const bf16_t*)qkv;
  p.out = (bf16_t*)out;
  p.lse2 = lse2;
  p.bias = bias;
  p.B = B; p.N = N; p.h = heads;
  p.scale = scale;
  p.sl2 = scale * LOG2E;
  hipStream_t st = (hipStream_t)stream;
  if (bias != nullptr) {
    if (hd == 128) return launch_fwd<128, true>(p, st);
    if (hd == 64) return launch_fwd<64, true>(p, st);
    return launch_fwd<32, true>(p, st);
  }
  if (hd == 128) return launch_fwd<128, false>(p, st);
  if (hd == 64) return launch_fwd<64, false>(p, st);
  return launch_fwd<32, false>(p, st);
}

extern "C" int htrvt_attn_bwd(const void* qkv, const float* bias, const void* out, const void* dout, const float* lse2,
                              float* delta, void* dqkv, float* dbias, int B, int N, int heads, int hd, float scale, int dtype,
                              void* stream) {
  HTRVT_REQUIRE(qkv && out && dout && lse2 && delta && dqkv, "htrvt_attn_bwd: null operand");
  HTRVT_REQUIRE(B > 0 && heads > 0 && htrvt_attn_supported(N, hd, dtype),
                "htrvt_attn_bwd: unsupported shape/dtype (N=%d >= 32, hd=%d in {32,64,128}, bfloat16)", N, hd);
  HTRVT_REQUIRE(bias == nullptr || N % 128 == 0, "htrvt_attn_bwd: with a score bias N=%d must be a multiple of 128", N);
  AttnParams p{};
  p.qkv = (const bf16_t*)qkv;
  p.o = (const bf16_t*)out;
  p.dout = (const bf16_t*)dout;
  p.lse2 = const_cast<float*>(lse2);
  p.dqkv = (bf16_t*)dqkv;
  p.bias = bias;
  p.dbias = dbias;
  p.B = B; p.N = N; p.h = heads;
  p.scale = scale;
  p.sl2 = scale * LOG2E;
  HTRVT_REQUIRE(dbias == nullptr || bias != nullptr, "htrvt_attn_bwd: dbias without a bias");   // bias without dbias: a constant mask
  hipStream_t st = (hipStream_t)stream;
  if (bias != nullptr) {
    if (hd == 128) return launch_bwd<128, true>(p, delta, st);
    if (hd == 64) return launch_bwd<64, true>(p, delta, st);
    return launch_bwd<32, true>(p, delta, st);
  }
  if (hd == 128) return launch_bwd<128, false>(p, delta, st);
  if (hd == 64) return launch_bwd<64, false>(p, delta, st);
  return launch_bwd<32, false>(p, delta, st);
}
