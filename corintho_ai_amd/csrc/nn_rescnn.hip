// nn_rescnn.hip -- K6: the north-star policy/value network, a 4-block residual
// CNN over the 4x4 board (specification: corintho_ai_amd/nets.py "rescnn4"), as
// ONE fused gfx950 kernel on fp32 MFMA.
//
// Mapping.  A 4x4 board has exactly 16 pixels -- one N-tile of
// v_mfma_f32_16x16x4_f32.  A 3x3 convolution is evaluated transposed, as for the
// MLP (nn_mlp.hip): out^T[co][pixel] += W[tap][ci][co] * in[ci][pixel + tap], so
//   A operand = a 16(co) x 4(ci) weight fragment,
//   B operand = the activation tile of ONE position shifted by the tap.
// The activation tile of a position lives in registers in accumulator layout
// (pixel on the lane, lane & 15; channel 16t + 4q + r in register r of tile t,
// q = lane >> 4), which IS the B layout when the K steps run in the order
// (tap, t, r) with k-slot q.  The spatial shift of a tap is a DPP row shift inside
// each 16-lane row (source pixel p + 4dy + dx, zero outside the row) plus a lane
// mask for the x wrap-around: "im2col" costs two VALU ops per B operand and no
// memory traffic at all.  Activations, the residual skip and the accumulators
// never leave registers through stem + 8 convolutions; bias + BatchNorm + ReLU +
// residual add run on the accumulators; the heads (1x1 convs, dense layers, tanh,
// 96-way softmax) are fused behind them.  Only weights move: one tap of one
// convolution (16 KB) at a time through a double-buffered LDS window filled by
// LDS-DMA (global_load_lds) while the previous tap computes.
//
// fp32 end to end, fixed k order, one position per MFMA column: a row's result
// does not depend on its batch (SURVEY 8e invariant).
// Work: 9.65 MFLOP per position, no padding waste in the 64->64 convolutions.
// Geometry: 256 threads = 4 waves, 4 positions per wave (every weight fragment
// feeds 16 MFMAs), 16 positions per workgroup.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "engine_defs.h"
#include "lds_dma.h"
#include "nn.h"

#define RC_NB 4
#define RC_POS_PER_WG 16
#define RC_STEM_CHUNK 1024   /* floats: 4 steps x 4 out tiles x 64 lanes */
#define RC_CONV_CHUNK 4096   /* floats: 16 steps x 4 out tiles x 64 lanes */
#define RC_NUM_CONVS 9       /* stem + 8 */
#define RC_NUM_CHUNKS 81
#define RC_TRUNK_FLOATS (9 * RC_STEM_CHUNK + 72 * RC_CONV_CHUNK)
#define RC_NUM_WEIGHTS 312383

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RcParams {
  const float *in;        /* [rows][80] */
  const int32_t *d_rows;
  const float *wtrunk;    /* RC_TRUNK_FLOATS, fragment order */
  const float *epi;       /* [9][3][64]: bias, bn scale, bn shift */
  const float *whead;     /* [16][64]   1x1 convs: rows 0..3 policy, 4..5 value */
  const float *head_epi;  /* [3][16] */
  const float *wpol;      /* [16][6][64] */
  const float *bpol;      /* [96] */
  const float *wv1;       /* [8][4][64] */
  const float *bv1;       /* [64] */
  const float *wv2;       /* [16][64] */
  const float *bv2;       /* [1] */
  float *eval;
  float *probs;
  CoNetIO io;             /* row indirection (evaluation cache), see nn.h */
};

/* input row / output element of launch row `pos` */
__device__ __forceinline__ size_t rc_in_row(const RcParams &P, int pos) { return (size_t)(P.io.in_idx ? P.io.in_idx[pos] : pos); }
__device__ __forceinline__ size_t rc_out_row(const RcParams &P, int pos) { return (size_t)(P.io.out_idx ? P.io.out_idx[pos] : pos); }

__device__ __forceinline__ const float *rc_chunk_ptr(const float *wtrunk, int ch) {
  return ch < 9 ? wtrunk + ch * RC_STEM_CHUNK : wtrunk + 9 * RC_STEM_CHUNK + (ch - 9) * RC_CONV_CHUNK;
}

/* LDS-DMA: each wave-instruction moves 1 KiB (lane i: bytes [16 i, 16 i + 16)) */
__device__ __forceinline__ void rc_stage(const float *wtrunk, float *lds_buf, int ch, int wave, int lane) {
  const float *src = rc_chunk_ptr(wtrunk, ch);
  const int pieces = ch < 9 ? RC_STEM_CHUNK / 256 : RC_CONV_CHUNK / 256;
  for (int p = wave; p < pieces; p += 4) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + p * 256 + lane * 4),
                                     (void __attribute__((address_space(3))) *)(lds_buf + p * 256), 16, 0, 0);
  }
}

/* the tap's view of an activation register: pixel p reads pixel p + S (S = 4 dy + dx),
 * zero outside the board */
template <int S>
__device__ __forceinline__ float rc_row_shift(float v) {
  if (S == 0) return v;
  constexpr int ctrl = S > 0 ? (0x100 + S) : (0x110 - S); /* row_shl:S reads lane+S, row_shr:S reads lane-S */
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), ctrl, 0xF, 0xF, true));
}

template <int TAP>
__device__ __forceinline__ float rc_tap(float v, bool okL, bool okR) {
  constexpr int dy = TAP / 3 - 1, dx = TAP % 3 - 1;
  float s = rc_row_shift<4 * dy + dx>(v);
  if (dx == -1) s = okL ? s : 0.0f;
  if (dx == 1) s = okR ? s : 0.0f;
  return s;
}

template <int CT, int TAP>
__device__ __forceinline__ void rc_conv_tap(f32x4 (&acc)[RC_NB][4], const float (&in)[RC_NB][4][4], const float *w, int lane,
                                            bool okL, bool okR) {
#pragma unroll
  for (int t = 0; t < CT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float *wp = w + ((t * 4 + r) * 4) * 64 + lane;
      float a[4];
#pragma unroll
      for (int to = 0; to < 4; ++to) a[to] = wp[to * 64];
#pragma unroll
      for (int nb = 0; nb < RC_NB; ++nb) {
        const float b = rc_tap<TAP>(in[nb][t][r], okL, okR);
#pragma unroll
        for (int to = 0; to < 4; ++to) acc[nb][to] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[to], b, acc[nb][to], 0, 0, 0);
      }
    }
  }
}

/* one 3x3 convolution = 9 weight chunks; chunk `ch` is consumed from LDS buffer ch & 1
 * while chunk ch + 1 streams into the other */
template <int CT>
__device__ __forceinline__ void rc_conv3x3(f32x4 (&acc)[RC_NB][4], const float (&in)[RC_NB][4][4], int &ch,
                                           const float *wtrunk, float *lds_w, int wave, int lane, bool okL, bool okR) {
#pragma unroll
  for (int nb = 0; nb < RC_NB; ++nb)
#pragma unroll
    for (int to = 0; to < 4; ++to) acc[nb][to] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define RC_TAP(T)                                                                         \
  {                                                                                       \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
    __syncthreads();                                                                      \
    if (ch + 1 < RC_NUM_CHUNKS) rc_stage(wtrunk, lds_w + ((ch + 1) & 1) * RC_CONV_CHUNK, ch + 1, wave, lane); \
    rc_conv_tap<CT, T>(acc, in, lds_w + (ch & 1) * RC_CONV_CHUNK, lane, okL, okR);        \
    ++ch;                                                                                 \
  }
  RC_TAP(0) RC_TAP(1) RC_TAP(2) RC_TAP(3) RC_TAP(4) RC_TAP(5) RC_TAP(6) RC_TAP(7) RC_TAP(8)
#undef RC_TAP
}

/* conv bias -> BatchNorm affine (-> + skip) -> ReLU, channel 16 to + 4 q + r */
template <bool ADD_SKIP, bool RELU>
__device__ __forceinline__ void rc_epilogue(float (&out)[RC_NB][4][4], const f32x4 (&acc)[RC_NB][4],
                                            const float (&skip)[RC_NB][4][4], const float *epi, int q) {
#pragma unroll
  for (int to = 0; to < 4; ++to) {
    const float4 b4 = *reinterpret_cast<const float4 *>(epi + 16 * to + 4 * q);
    const float4 a4 = *reinterpret_cast<const float4 *>(epi + 64 + 16 * to + 4 * q);
    const float4 c4 = *reinterpret_cast<const float4 *>(epi + 128 + 16 * to + 4 * q);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
    const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
    const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
    for (int nb = 0; nb < RC_NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[nb][to][r] + bb[r];
        v = aa[r] * v + cc[r];
        if (ADD_SKIP) v = skip[nb][to][r] + v;
        if (RELU) v = v > 0.0f ? v : 0.0f;
        out[nb][to][r] = v;
      }
  }
}

/* ---- heads, shared by both precisions: x = the trunk output of this wave's RC_NB
 * positions (fp32, accumulator layout); feat_w = RC_NB x 96 floats of LDS owned by the wave */
__device__ __forceinline__ void rc_dense_policy(const RcParams &P, const float *wpol, const float *feat16, int rows,
                                                int pos_base, int lane, int ncols = 16);
__device__ __forceinline__ void rc_dense_value(const RcParams &P, const float *wv1, const float *wv2, const float *feat16,
                                               int rows, int pos_base, int lane, int ncols = 16);

__device__ __forceinline__ void rc_heads(const RcParams &P, const float (&x)[RC_NB][4][4], float *feat_wg, int wave,
                                         int rows, int row0, int lane, int q, int c) {
  float *feat_w = feat_wg + wave * RC_NB * 96;
  /* ---- heads.  1x1 convolutions: out rows 0..3 policy planes, 4..5 value planes */
  f32x4 h1[RC_NB];
#pragma unroll
  for (int nb = 0; nb < RC_NB; ++nb) h1[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = P.whead[(t * 4 + r) * 64 + lane];
#pragma unroll
      for (int nb = 0; nb < RC_NB; ++nb) h1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x[nb][t][r], h1[nb], 0, 0, 0);
    }
  {
    const float4 b4 = *reinterpret_cast<const float4 *>(P.head_epi + 4 * q);
    const float4 a4 = *reinterpret_cast<const float4 *>(P.head_epi + 16 + 4 * q);
    const float4 c4 = *reinterpret_cast<const float4 *>(P.head_epi + 32 + 4 * q);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
    const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
    const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
    for (int nb = 0; nb < RC_NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = h1[nb][r] + bb[r];
        v = aa[r] * v + cc[r];
        v = v > 0.0f ? v : 0.0f;
        /* flatten: policy index pixel*4 + ch, value index 64 + pixel*2 + ch */
        if (q == 0) feat_w[nb * 96 + c * 4 + r] = v;
        if (q == 1 && r < 2) feat_w[nb * 96 + 64 + c * 2 + r] = v;
      }
  }
  __syncthreads();
  /* the workgroup's 16 positions = one column tile: wave 0 policy, wave 1 value */
  if (wave == 0) rc_dense_policy(P, P.wpol, feat_wg, rows, row0, lane);
  if (wave == 1) rc_dense_value(P, P.wv1, P.wv2, feat_wg, rows, row0, lane);
}

/* Dense heads on 16 positions at once (one full MFMA column tile): `feat` = 16 x 96 floats of
 * LDS (flattened head features of consecutive positions pos_base .. pos_base + 15), column c =
 * position.  One wave runs the policy head (Dense 64 -> 96, softmax) of a tile, another its
 * value head (Dense 32 -> 64, ReLU, Dense 64 -> 1, tanh), so the dense weights are read once
 * per 16 positions and no MFMA column is idle. */
__device__ __forceinline__ void rc_dense_policy(const RcParams &P, const float *wpol, const float *feat16, int rows,
                                                int pos_base, int lane, int ncols) {
  const int q = lane >> 4, c = lane & 15;
  const float *feat = feat16 + c * 96;
  f32x4 pl[6];
#pragma unroll
  for (int to = 0; to < 6; ++to) pl[to] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float b = feat[4 * s + q];
#pragma unroll
    for (int to = 0; to < 6; ++to)
      pl[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(wpol[(s * 6 + to) * 64 + lane], b, pl[to], 0, 0, 0);
  }
  /* softmax over the 96 logits of column c: registers (to, r) in the lane, q across lanes */
  float lg[6][4];
  float m = -INFINITY;
#pragma unroll
  for (int to = 0; to < 6; ++to) {
    const float4 b4 = *reinterpret_cast<const float4 *>(P.bpol + 16 * to + 4 * q);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lg[to][r] = pl[to][r] + bb[r];
      m = lg[to][r] > m ? lg[to][r] : m;
    }
  }
  float o = __shfl_xor(m, 16, 64);
  m = o > m ? o : m;
  o = __shfl_xor(m, 32, 64);
  m = o > m ? o : m;
  float sum = 0.0f;
#pragma unroll
  for (int to = 0; to < 6; ++to)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      /* exp(x) = 2^(x log2 e) on the hardware exponential (1 ulp; x <= 0 here) */
      lg[to][r] = __builtin_amdgcn_exp2f((lg[to][r] - m) * 1.44269504088896340736f);
      sum += lg[to][r];
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  const int pos = pos_base + c;
  if (pos < rows && c < ncols) { /* a workgroup of the thin kernel owns the first ncols = 8 columns of the tile only */
    const size_t orow = rc_out_row(P, pos);
#pragma unroll
    for (int to = 0; to < 6; ++to) {
      float4 p = make_float4(lg[to][0] * inv, lg[to][1] * inv, lg[to][2] * inv, lg[to][3] * inv);
      *reinterpret_cast<float4 *>(P.probs + orow * (size_t)P.io.probs_stride + 16 * to + 4 * q) = p;
    }
  }
}

__device__ __forceinline__ void rc_dense_value(const RcParams &P, const float *wv1, const float *wv2, const float *feat16,
                                               int rows, int pos_base, int lane, int ncols) {
  const int q = lane >> 4, c = lane & 15;
  const float *feat = feat16 + c * 96;
  f32x4 v1[4];
#pragma unroll
  for (int to = 0; to < 4; ++to) v1[to] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float b = feat[64 + 4 * s + q];
#pragma unroll
    for (int to = 0; to < 4; ++to)
      v1[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv1[(s * 4 + to) * 64 + lane], b, v1[to], 0, 0, 0);
  }
  f32x4 v2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 b4 = *reinterpret_cast<const float4 *>(P.bv1 + 16 * t + 4 * q);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float hv = v1[t][r] + bb[r];
      hv = hv > 0.0f ? hv : 0.0f;
      v2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv2[(t * 4 + r) * 64 + lane], hv, v2, 0, 0, 0);
    }
  }
  const int pos = pos_base + c;
  if (q == 0 && pos < rows && c < ncols) P.eval[rc_out_row(P, pos) * (size_t)P.io.eval_stride] = tanhf(v2[0] + P.bv2[0]);
}

#ifndef CO_RC_F32_BLOCKS
#define CO_RC_F32_BLOCKS 2
#endif
__global__ __launch_bounds__(256, CO_RC_F32_BLOCKS) void co_k_rescnn_forward(RcParams P) {
  __shared__ __attribute__((aligned(16))) float lds_w[2 * RC_CONV_CHUNK];
  __shared__ float lds_feat[4][RC_NB][96];
  const int rows = *P.d_rows;
  const int row0 = blockIdx.x * RC_POS_PER_WG;
  if (row0 >= rows) return;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, c = lane & 15;
  const bool okL = (c & 3) != 0, okR = (c & 3) != 3;
  rc_stage(P.wtrunk, lds_w, 0, wave, lane);

  /* input planes: lane (q, pixel c) holds channels 4q..4q+3 -- q 0: the cell's four
   * board bits, q 1: reserves 0..3, q 2: reserves 4..5 (+ zero padding), q 3: zeros */
  float x[RC_NB][4][4];
#pragma unroll
  for (int nb = 0; nb < RC_NB; ++nb) {
    const int pos = row0 + wave * RC_NB + nb;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pos < rows) v = *reinterpret_cast<const float4 *>(P.in + rc_in_row(P, pos) * CO_STATE_STRIDE + (q == 0 ? 4 * c : 60 + 4 * q));
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[nb][t][r] = 0.0f;
    x[nb][0][0] = v.x;
    x[nb][0][1] = v.y;
    x[nb][0][2] = v.z;
    x[nb][0][3] = v.w;
  }

  f32x4 acc[RC_NB][4];
  float y[RC_NB][4][4];
  int ch = 0;
  /* stem */
  rc_conv3x3<1>(acc, x, ch, P.wtrunk, lds_w, wave, lane, okL, okR);
  rc_epilogue<false, true>(x, acc, x, P.epi, q);
  /* residual tower */
  for (int b = 0; b < 4; ++b) {
    rc_conv3x3<4>(acc, x, ch, P.wtrunk, lds_w, wave, lane, okL, okR);
    rc_epilogue<false, true>(y, acc, x, P.epi + (size_t)(1 + 2 * b) * 192, q);
    rc_conv3x3<4>(acc, y, ch, P.wtrunk, lds_w, wave, lane, okL, okR);
    rc_epilogue<true, true>(x, acc, x, P.epi + (size_t)(2 + 2 * b) * 192, q);
  }

  rc_heads(P, x, &lds_feat[0][0][0], wave, rows, row0, lane, q, c);
}

/* ======================================================================
 * Split-precision variants (CO_NET_RESCNN4_X3: NT = 2 terms, CO_NET_RESCNN4_X6: NT = 3 terms):
 * same network, same weights, same register-resident structure, but every 3x3 convolution runs
 * on the bf16 matrix pipe with both operands written as a sum of NT bf16 values,
 *   x = x0 + x1 (+ x2),  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1),
 * and the product expanded into the terms w_i x_j with i + j <= NT - 1 (fp32 accumulation in the
 * MFMA):
 *   NT = 2 ("bf16x3"): 16 significand bits kept, 3 MFMAs; dropped terms ~2^-16 |x w|.  Within
 *           2e-5 of the float32 restatement -- narrower than the reference's float32 arithmetic.
 *   NT = 3 ("bf16x6"): x0 + x1 + x2 IS the float32 value (3 x 8 = 24 significand bits, the
 *           remainders are exact), 6 MFMAs; the dropped terms w1 x2, w2 x1, w2 x2 are <= 2^-24
 *           |x w| each -- the size of ONE float32 rounding of the product, and there are fewer
 *           accumulator roundings than in the fp32 MFMA chain (one per 16 products instead of one
 *           per product).  Measured against a float64 restatement the error is that of K6 (fp32
 *           MFMA) or smaller (tests/test_net_precision.py): float32-equivalent arithmetic at 16/6 =
 *           2.7x the fp32 matrix rate.
 *
 * v_mfma_f32_32x32x16_bf16 (an MFMA of this shape occupies the SIMD's issue port for 8 of
 * its 32 cycles; the 16x16x32 shape for 8 of 16, which left too little room for the DPP
 * shifts).  Its 32 columns are TWO positions (lane & 31 = position*16 + pixel), its 32 rows
 * half of the 64 output channels.  A lane (h = lane >> 5) owns 16 channels of each row tile
 * T: channel 32T + 4h + 8g + i in accumulator register 4g + i.  One K step = 16 input
 * channels = the lane's registers 8a..8a+7 of tile T (k-slot (h, j) <-> channel
 * 32T + 4h + 8(2a + j/4) + j%4), packed two bf16 per VGPR: again the output layout of one
 * layer is the operand layout of the next, and the tap shift is the same DPP row shift
 * (a row of 16 lanes = one position), now on packed pairs.
 * Bias, BatchNorm, residual adds and the heads stay in fp32.
 * Geometry: 512 threads = 8 waves (two per SIMD), NP position pairs per wave. */
/* NP = position pairs per wave.  NT = 2: 2 in the throughput kernel (32 positions per workgroup), 1 in
 * the small-batch kernel (16 per workgroup: half the MFMA work behind the same weight stream, so a
 * batch that fits one round of workgroups comes back sooner -- the thinning tail of a generation
 * runs hundreds of such iterations, each as long as its slowest kernel).  NT = 3: 1 (three packed
 * operand sets + three weight fragment sets leave no registers for a second pair at two waves per
 * SIMD; the MFMA work per weight byte is that of NT = 2, NP = 2 again). */
#ifndef RC3_SMALL_ROWS
#define RC3_SMALL_ROWS 4096 /* NT = 2: batches up to this size take the small-batch kernel: <= 256 workgroups */
#endif
#define RC6_THIN_ROWS 2048  /* NT = 3: batches up to this size take the four-wave kernel: <= 256 workgroups of 8 positions */
#define RCS_STEM_CHUNK(NT) (512 * (NT))  /* u32: 1 k-step x 2 out tiles x NT terms x 64 lanes x 4 */
#define RCS_CONV_CHUNK(NT) (2048 * (NT)) /* u32: 4 k-steps ... = 8 KB per term */
#define RCS_TRUNK_WORDS(NT) (9 * RCS_STEM_CHUNK(NT) + 72 * RCS_CONV_CHUNK(NT))
#define RC3_EPI_WORDS 1792 /* 9 convolutions x (bias, BN scale, BN shift)[64], padded to whole 256-word pieces */
/* head weights: 1x1 fragments (4 steps x NT terms x 64 lanes x 4 words), then the fp32 dense weights in
 * MFMA order: policy dense (6144), value dense 1 (2048), value dense 2 (1024) */
#define RCS_FRAG1_WORDS(NT) (4 * (NT) * 256)
#define RCS_DENSE_WORDS (6144 + 2048 + 1024)
#define RCS_HEAD_WORDS(NT) (RCS_FRAG1_WORDS(NT) + RCS_DENSE_WORDS)
/* weights stream through LDS in groups of three taps (one kernel row): 27 groups, group
 * gi < 3 belongs to the stem */
#define RC3_NUM_GROUPS 27
#define RCS_GROUP_WORDS(NT) (3 * RCS_CONV_CHUNK(NT)) /* 48 KB / 72 KB */
#define RCS_FEAT_WORDS(NP) (8 * 2 * (NP) * 96)
/* LDS: [2 weight groups][NT = 2: head features][epilogue constants][NT = 2: head weights].  With three
 * terms the head weights do not fit beside two 72 KB groups: they are staged into the idle group
 * buffer while the last group computes, and the head features go where the last group was once every
 * wave has left it.  151 KB: what is left of the CU's 160 KB (and of its registers, see the kernel's
 * attributes) is room for wavefronts of the search kernel beside this one. */
#define RCS_LDS_WORDS(NT, NP) \
  (2 * RCS_GROUP_WORDS(NT) + ((NT) == 2 ? RCS_FEAT_WORDS(NP) : 0) + RC3_EPI_WORDS + ((NT) == 2 ? RCS_HEAD_WORDS(NT) : 0))

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Rc3Params {
  RcParams base;          /* dense heads + epilogue parameters, in/out pointers */
  const uint32_t *wtrunk; /* RCS_TRUNK_WORDS, bf16 term fragments */
  const uint32_t *whead3; /* RCS_HEAD_WORDS: [4 steps][NT terms][64 lanes][4 words] 1x1 head convs in fragment order,
                           * then the fp32 dense weights wpol, wv1, wv2 as in RcParams */
  const uint32_t *epi3;   /* RC3_EPI_WORDS: RcParams::epi, padded */
  uint32_t *range_flag;   /* f16x3: raised when an activation beyond fp16's range was split (nn.h range_exceeded) */
  int32_t pass_rows;      /* f16x3 with the pixel-major kernel: rows of one pass of that kernel over the chip (32 per CU);
                           * 0 = batches are not split between the kernels (rcp_small_begin) */
};

/* Which rows of a batch does the small-batch kernel take?  Without the pixel-major kernel: all of a batch of up to
 * RC3_SMALL_ROWS rows, none of a larger one.  With it the batch is SPLIT on the device: the pixel-major kernel runs one
 * workgroup of 32 rows per CU and a pass costs its full time however few of its workgroups have rows, so it takes the
 * whole passes of the batch, plus a remainder of more than RC3_SMALL_ROWS rows; a smaller remainder -- half of all
 * batches -- goes to the small-batch kernel (16 rows per workgroup; 8 on its thin path), which is through in a third to two
 * thirds of a pass.  A row's result does not depend on the kernel that evaluates it.  -> the small kernel's first row
 * (Only for a launch that has the GPU to itself, CoNetIO::alone: measured in round 4, 10 000 / 12 288 / 20 000 rows alone
 * 0.185 / 0.217 / 0.345 ms against 0.27 / 0.26 / 0.40 unsplit -- but beside the other pool's kernels, where CUs and not
 * latency are scarce, the pixel-major kernel's 32 rows per 160 us of a CU beat the small kernel's 16 per 110: a
 * two-pool generation 438.7 ms split against 434.0 unsplit.) */
__device__ __forceinline__ int rcp_small_begin(const Rc3Params &Q, int rows) {
  /* (a batch the host queues no throughput kernel for -- rows_cap <= RC3_SMALL_ROWS -- is the small kernel's whole,
   * whatever a pass is: on a device or partition of <= 128 CUs a pass is <= RC3_SMALL_ROWS rows, and `full` below would
   * hand rows to a kernel that was never launched) */
  if (rows <= RC3_SMALL_ROWS) return 0;
  if (Q.pass_rows <= 0) return rows;
  const int full = rows / Q.pass_rows * Q.pass_rows;
  return rows - full > RC3_SMALL_ROWS ? rows : full;
}

template <int NT>
__device__ __forceinline__ const uint32_t *rcs_group_ptr(const uint32_t *wtrunk, int gi) {
  return gi < 3 ? wtrunk + gi * 3 * RCS_STEM_CHUNK(NT) : wtrunk + 9 * RCS_STEM_CHUNK(NT) + (gi - 3) * 3 * RCS_CONV_CHUNK(NT);
}

/* LDS-DMA of `words` (a multiple of 256) by the eight waves of the workgroup (lds_dma.h: the waits
 * are the kernel's own) */
__device__ __forceinline__ void rcs_stage_words(const uint32_t *src, uint32_t lds_addr, int words, int wave, int lane,
                                                int nw = 8) {
  const int pieces = words / 256;
  for (int p = wave; p < pieces; p += nw) co_lds_dma_1k(src + p * 256 + lane * 4, lds_addr + (uint32_t)p * 1024u);
}

template <int NT>
__device__ __forceinline__ void rcs_stage(const uint32_t *wtrunk, uint32_t lds_addr, int gi, int wave, int lane, int nw = 8) {
  rcs_stage_words(rcs_group_ptr<NT>(wtrunk, gi), lds_addr, gi < 3 ? 3 * RCS_STEM_CHUNK(NT) : 3 * RCS_CONV_CHUNK(NT), wave, lane, nw);
}

/* (a, b) -> NT packed 16-bit pairs: the values rounded to bf16 (F16: to fp16), then the successive remainders
 * (each remainder is exact in float32, so with three bf16 terms they add up to the float32 value; two fp16
 * terms keep 2 x 11 = 22 significand bits) */
template <int NT, bool F16 = false>
__device__ __forceinline__ void rcs_split(float a, float b, uint32_t (&t)[NT]) {
  if constexpr (F16 && NT == 2) {
    /* three instructions instead of five: the pair's first terms, then each second term as ONE mixed-precision fma,
     * f16(a - float(t0.lo)) -- the difference is exact in float32 (see above), so the one rounding is the conversion's,
     * as before: the same bits.  (An epilogue of the f16x3 kernels is vector-issue-bound: 336 -> 272 instructions per
     * wave in the pixel-major kernel; round 5.) */
    const f32x2 v2 = {a, b};
    const uint32_t t0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, f16x2));
    uint32_t t1;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(t1)
        : "v"(t0), "v"(a), "v"(b));
    t[0] = t0;
    t[1] = t1;
    return;
  }
  f32x2 v = {a, b};
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if constexpr (F16) {
      f16x2 hb = __builtin_convertvector(v, f16x2);
      t[i] = __builtin_bit_cast(uint32_t, hb);
      if (i + 1 < NT) {
        f32x2 hf = __builtin_convertvector(hb, f32x2);
        v = (f32x2){v.x - hf.x, v.y - hf.y};
      }
    } else {
      bf16x2 hb = __builtin_convertvector(v, bf16x2);
      t[i] = __builtin_bit_cast(uint32_t, hb);
      if (i + 1 < NT) {
        f32x2 hf = __builtin_convertvector(hb, f32x2);
        v = (f32x2){v.x - hf.x, v.y - hf.y};
      }
    }
  }
}

/* The range guard of the f16x3 kinds (nn.h range_exceeded) follows the FIRST terms as they are split: the running maximum
 * of the packed fp16 pairs, one v_pk_max_f16 per pair of activations (on the float32 values it was two v_max_f32 per pair
 * in a vector-issue-bound epilogue).  What is split is an input plane or the output of a ReLU, never negative; an
 * activation beyond fp16's range has the first term +inf -- exactly the event the guard reports (a NaN can only follow an
 * infinity, which is reported when it appears). */
__device__ __forceinline__ uint32_t rcs_pk_max_f16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ bool rcs_pk_f16_finite(uint32_t pk) { return (pk & 0x7FFFu) < 0x7C00u && ((pk >> 16) & 0x7FFFu) < 0x7C00u; }

/* one v_mfma_f32_32x32x16 on packed 16-bit operands: bf16 terms, or fp16 terms */
template <bool F16>
__device__ __forceinline__ f32x16 rcs_mfma(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int S>
__device__ __forceinline__ uint32_t rc3_row_shift(uint32_t v) {
  if (S == 0) return v;
  constexpr int ctrl = S > 0 ? (0x100 + S) : (0x110 - S);
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, ctrl, 0xF, 0xF, true);
}

/* The four packed words of one B operand shifted to tap (dy, dx).  For dx != 0 the shift and
 * the zeroing of the lanes whose source pixel lies in the neighbouring board row are ONE
 * instruction, v_cndmask_b32 with a DPP source: D = vcc ? 0 : row_shift(v) with vcc = the wrap
 * lanes (x = 0 for dx = -1, x = 3 for dx = +1; a constant lane pattern).  As two instructions
 * (v_mov_b32_dpp + v_cndmask_b32_e64) the B-operand preparation took 2.7 vector issues per MFMA
 * and, with two waves per SIMD, left the issue port nearly full.  The trailing s_nop 1 covers
 * the VALU-write -> MFMA-read wait states that the compiler cannot see into the asm for. */
#define RC3_CNDMASK_DPP4(CTRL)                                                                          \
  asm("s_mov_b64 vcc, %[m]\n\t"                                                                         \
      "v_cndmask_b32_dpp %[o0], %[i0], %[z], vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
      "v_cndmask_b32_dpp %[o1], %[i1], %[z], vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
      "v_cndmask_b32_dpp %[o2], %[i2], %[z], vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
      "v_cndmask_b32_dpp %[o3], %[i3], %[z], vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   \
      "s_nop 1"                                                                                         \
      : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3)                                  \
      : [i0] "v"(in[0]), [i1] "v"(in[1]), [i2] "v"(in[2]), [i3] "v"(in[3]), [z] "v"(zero), [m] "s"(wrap) \
      : "vcc")

template <int TAP>
__device__ __forceinline__ u32x4 rc3_tap4(const uint32_t (&in)[4], uint32_t zero) {
  constexpr int dy = TAP / 3 - 1, dx = TAP % 3 - 1;
  u32x4 out;
  if constexpr (dx == 0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) out[m] = rc3_row_shift<4 * dy>(in[m]);
  } else {
    const unsigned long long wrap = dx < 0 ? 0x1111111111111111ull : 0x8888888888888888ull;
    uint32_t o0, o1, o2, o3;
    if constexpr (4 * dy + dx == -5) RC3_CNDMASK_DPP4("row_shr:5");
    if constexpr (4 * dy + dx == -3) RC3_CNDMASK_DPP4("row_shr:3");
    if constexpr (4 * dy + dx == -1) RC3_CNDMASK_DPP4("row_shr:1");
    if constexpr (4 * dy + dx == 1) RC3_CNDMASK_DPP4("row_shl:1");
    if constexpr (4 * dy + dx == 3) RC3_CNDMASK_DPP4("row_shl:3");
    if constexpr (4 * dy + dx == 5) RC3_CNDMASK_DPP4("row_shl:5");
    out[0] = o0;
    out[1] = o1;
    out[2] = o2;
    out[3] = o3;
  }
  return out;
}

/* tap as a value: after full unrolling every call site has a constant tap and folds to one case */
__device__ __forceinline__ u32x4 rc3_tap4_sel(const uint32_t (&in)[4], int tap, uint32_t zero) {
  switch (tap) {
    case 0: return rc3_tap4<0>(in, zero);
    case 1: return rc3_tap4<1>(in, zero);
    case 2: return rc3_tap4<2>(in, zero);
    case 3: return rc3_tap4<3>(in, zero);
    case 4: return rc3_tap4<4>(in, zero);
    case 5: return rc3_tap4<5>(in, zero);
    case 6: return rc3_tap4<6>(in, zero);
    case 7: return rc3_tap4<7>(in, zero);
    default: return rc3_tap4<8>(in, zero);
  }
}

/* One staged group = taps 3G .. 3G + 2, CS K steps each.  p[t][np][s][m]: term t, K step s = 2T + a,
 * word m = channels (reg 8a + 2m, 8a + 2m + 1) of tile T.  The weight fragments of step i + 1
 * (also across the tap boundary) are requested from LDS before the MFMAs of step i issue (two
 * register sets), so the LDS latency is paid once per group.  Products w_i x_j, i + j <= NT - 1,
 * largest first. */
template <int CS, int G, int NP, int NT, bool F16 = false>
__device__ __forceinline__ void rcs_conv_group(f32x16 (&acc)[NP][2], const uint32_t (&p)[NT][NP][4][4], const uint32_t *wg,
                                               int lane) {
  /* terms the B operand has: the stem's inputs (board bits 0 / 1, reserves k / 4) are exact in bf16 */
  constexpr int XT = CS == 1 ? 1 : NT;
  constexpr int tw = CS == 1 ? RCS_STEM_CHUNK(NT) : RCS_CONV_CHUNK(NT);
  constexpr int N = 3 * CS;
  uint32_t zero;
  asm("v_mov_b32 %0, 0" : "=v"(zero)); /* a zero the compiler keeps in a VGPR (second cndmask source) */
  u32x4 a[2][NT][2];
#pragma unroll
  for (int to = 0; to < 2; ++to)
#pragma unroll
    for (int t = 0; t < NT; ++t) a[0][t][to] = *reinterpret_cast<const u32x4 *>(wg + (((0 * 2 + to) * NT + t) * 64 + lane) * 4);
#pragma unroll
  for (int idx = 0; idx < N; ++idx) {
    const int cur = idx & 1, nxt = cur ^ 1;
    const int tg = idx / CS, s = idx % CS;
    if (idx + 1 < N) {
      const int tg1 = (idx + 1) / CS, s1 = (idx + 1) % CS;
      const uint32_t *w1 = wg + tg1 * tw;
#pragma unroll
      for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          a[nxt][t][to] = *reinterpret_cast<const u32x4 *>(w1 + (((s1 * 2 + to) * NT + t) * 64 + lane) * 4);
    }
#pragma unroll
    for (int np = 0; np < NP; ++np) {
      u32x4 B[XT];
#pragma unroll
      for (int t = 0; t < XT; ++t) B[t] = rc3_tap4_sel(p[t][np][s], 3 * G + tg, zero);
#pragma unroll
      for (int sum = 0; sum < NT; ++sum)
#pragma unroll
        for (int i = 0; i <= sum; ++i)
          if (sum - i < XT) {
#pragma unroll
            for (int to = 0; to < 2; ++to)
              acc[np][to] = rcs_mfma<F16>(a[cur][i][to], B[sum - i], acc[np][to]);
          }
    }
  }
}

template <int CS, int NP, int NT, int NW, bool F16 = false>
__device__ __forceinline__ void rcs_conv3x3(f32x16 (&acc)[NP][2], const uint32_t (&p)[NT][NP][4][4], int &ch,
                                            const Rc3Params &Q, uint32_t *lds_w, uint32_t lds_w_addr, int wave, int lane) {
#pragma unroll
  for (int np = 0; np < NP; ++np)
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[np][to][i] = 0.0f;
#define RC3_GROUP(G)                                                                                              \
  {                                                                                                               \
    CO_WAIT_VMCNT(0); /* group ch has landed (requested one group ago) */                                         \
    co_wg_barrier();  /* ... for every wave, and everyone has left the other buffer */                            \
    if (ch + 1 < RC3_NUM_GROUPS)                                                                                  \
      rcs_stage<NT>(Q.wtrunk, lds_w_addr + (uint32_t)((ch + 1) & 1) * (RCS_GROUP_WORDS(NT) * 4u), ch + 1, wave, lane, NW); \
    else if (NT != 2) /* the head weights ride in the buffer the last group leaves idle */                        \
      rcs_stage_words(Q.whead3, lds_w_addr + (uint32_t)((ch + 1) & 1) * (RCS_GROUP_WORDS(NT) * 4u), RCS_HEAD_WORDS(NT), wave, lane, NW); \
    rcs_conv_group<CS, G, NP, NT, F16>(acc, p, lds_w + (ch & 1) * RCS_GROUP_WORDS(NT), lane);                          \
    ++ch;                                                                                                         \
  }
  RC3_GROUP(0) RC3_GROUP(1) RC3_GROUP(2)
#undef RC3_GROUP
}

/* fp32 tile values -> the packed operands of the next convolution */
template <int NP, int NT, bool F16 = false>
__device__ __forceinline__ void rcs_pack(uint32_t (&p)[NT][NP][4][4], const float (&v)[NP][2][16], uint32_t &amax) {
#pragma unroll
  for (int np = 0; np < NP; ++np)
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          uint32_t t[NT];
          /* (what is packed is an input plane or the output of a ReLU: never negative) */
          rcs_split<NT, F16>(v[np][T][8 * a + 2 * m], v[np][T][8 * a + 2 * m + 1], t);
          if constexpr (F16) amax = rcs_pk_max_f16(amax, t[0]);
#pragma unroll
          for (int i = 0; i < NT; ++i) p[i][np][2 * T + a][m] = t[i];
        }
}

/* conv bias -> BatchNorm affine (-> + skip) -> ReLU; register 4g + i of tile T is channel
 * 32T + 8g + 4h + i */
template <bool ADD_SKIP, int NP>
__device__ __forceinline__ void rc3_epilogue(float (&out)[NP][2][16], const f32x16 (&acc)[NP][2],
                                             const float (&skip)[NP][2][16], const float *epi, int h) {
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int chn = 32 * T + 8 * g + 4 * h;
      const float4 b4 = *reinterpret_cast<const float4 *>(epi + chn);
      const float4 a4 = *reinterpret_cast<const float4 *>(epi + 64 + chn);
      const float4 c4 = *reinterpret_cast<const float4 *>(epi + 128 + chn);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
      const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
      const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
      float cb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) cb[i] = __builtin_fmaf(aa[i], bb[i], cc[i]);
#pragma unroll
      for (int np = 0; np < NP; ++np)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          /* a (acc + bias) + c as one fma on the folded shift cb = a bias + c */
          float v = __builtin_fmaf(aa[i], acc[np][T][4 * g + i], cb[i]);
          if (ADD_SKIP) v = skip[np][T][4 * g + i] + v;
          v = v > 0.0f ? v : 0.0f;
          out[np][T][4 * g + i] = v;
        }
    }
}

#ifdef CO_PROF
/* diagnostic builds: cycles of wave 0 of every workgroup by phase (tools/prof_nn.py) */
__device__ unsigned long long rc3_prof[12]; /* 0..5 phases, 6 whole pass, 7 passes, 8 whole pass in 100 MHz ticks; K6p only: 9 waited for the
                                             * weight DMA, 10 waited at the tap barrier, 11 multiplied (inside phases 1 and 3) */
#define RC3_STAMP(slot)                                                              \
  {                                                                                  \
    unsigned long long now_ = __builtin_readcyclecounter();                          \
    if (tid == 0) atomicAdd(&rc3_prof[slot], now_ - stamp_);                         \
    stamp_ = now_;                                                                   \
  }
extern "C" int ca_net_prof(unsigned long long out[12]) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rc3_prof), sizeof(rc3_prof)) == hipSuccess ? 0 : 1;
}
/* K6p: the core-clock stamps of workgroup 0's eight waves at the nine tap barriers of ONE trunk convolution (the fifth
 * convolution of the kernel): [wave][tap][arrived, left], [wave][18] = the convolution's end (tools/prof_nn.py with NN_TRACE=1) */
__device__ unsigned rc3_trace[8 * 20];
extern "C" int ca_net_trace(unsigned out[160]) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rc3_trace), sizeof(rc3_trace)) == hipSuccess ? 0 : 1;
}
#else
#define RC3_STAMP(slot)
#endif

/* NW = waves per workgroup: 8 (two per SIMD), or 4 in the thin-batch kernel of NT = 3 (below) */
template <int NP, int NT, int NW = 8, bool F16 = false>
__device__ __forceinline__ void rcs_forward(const Rc3Params &Q, const int rbase = 0) {
  const RcParams &P = Q.base;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  uint32_t *lds_w = lds_dyn;
  /* NT = 3: the features reuse the buffer of the last group (RC3_NUM_GROUPS - 1 = 26 -> buffer 0), free behind
   * the barrier in front of the heads */
  float *lds_feat = reinterpret_cast<float *>(NT == 2 ? lds_dyn + 2 * RCS_GROUP_WORDS(NT) : lds_dyn);
  const int rows = *P.d_rows; /* this launch works on rows rbase .. rows - 1 */
  if (NT == 2 && (rows - rbase <= RC3_SMALL_ROWS) != (NP == 1)) return; /* the other kernel takes this batch */
  const int row0 = rbase + blockIdx.x * (2 * NP * NW);
  if (row0 >= rows) return;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, h = lane >> 5, p2 = (lane >> 4) & 1, c = lane & 15;
#ifdef CO_PROF
  unsigned long long stamp_ = __builtin_readcyclecounter();
  const unsigned long long start_ = stamp_, real_ = __builtin_amdgcn_s_memrealtime();
#endif
  const uint32_t lds_w_addr = co_lds_addr(lds_dyn);
  constexpr int epi_off = 2 * RCS_GROUP_WORDS(NT) + (NT == 2 ? RCS_FEAT_WORDS(NP) : 0);
  uint32_t *lds_epi_w = lds_dyn + epi_off;
  const uint32_t lds_epi_addr = lds_w_addr + epi_off * 4u;
  const uint32_t *lds_head = NT == 2 ? lds_epi_w + RC3_EPI_WORDS : lds_w + (RC3_NUM_GROUPS & 1) * RCS_GROUP_WORDS(NT);

  /* input planes: register 4g + i of tile 0 = channel 8g + 4h + i:
   * g 0: h 0 the cell's board bits, h 1 reserves 0..3; g 1: h 0 reserves 4..5 (+ padding), h 1 zeros */
  float x[NP][2][16];
#pragma unroll
  for (int np = 0; np < NP; ++np) {
    const int pos = row0 + wave * (2 * NP) + np * 2 + p2;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int i = 0; i < 16; ++i) x[np][T][i] = 0.0f;
    if (pos < rows) {
      const float *row = P.in + rc_in_row(P, pos) * CO_STATE_STRIDE;
      const float4 v0 = *reinterpret_cast<const float4 *>(row + (h == 0 ? 4 * c : 64));
      const float4 v1 = *reinterpret_cast<const float4 *>(row + (h == 0 ? 68 : 72));
      x[np][0][0] = v0.x; x[np][0][1] = v0.y; x[np][0][2] = v0.z; x[np][0][3] = v0.w;
      x[np][0][4] = v1.x; x[np][0][5] = v1.y; x[np][0][6] = v1.z; x[np][0][7] = v1.w;
    }
  }
  uint32_t pk[NT][NP][4][4];
  uint32_t amax = 0u; /* (packed fp16 pair: rcs_pk_max_f16) */
  rcs_pack<NP, NT, F16>(pk, x, amax);
  /* weight stream, requested behind the input loads (vmcnt retires in issue order): group 0, the
   * epilogue constants and, with two terms, the head weights (the wait before the first MFMA
   * covers them) */
  rcs_stage<NT>(Q.wtrunk, lds_w_addr, 0, wave, lane, NW);
  rcs_stage_words(Q.epi3, lds_epi_addr, RC3_EPI_WORDS, wave, lane, NW);
  if (NT == 2) rcs_stage_words(Q.whead3, lds_epi_addr + RC3_EPI_WORDS * 4u, RCS_HEAD_WORDS(NT), wave, lane, NW);
  RC3_STAMP(0)
  f32x16 acc[NP][2];
  float y[NP][2][16];
  int ch = 0;
  rcs_conv3x3<1, NP, NT, NW, F16>(acc, pk, ch, Q, lds_w, lds_w_addr, wave, lane);
  RC3_STAMP(1)
  const float *lds_epi = reinterpret_cast<const float *>(lds_epi_w);
  rc3_epilogue<false, NP>(x, acc, x, lds_epi, h);
  rcs_pack<NP, NT, F16>(pk, x, amax);
  RC3_STAMP(2)
  for (int b = 0; b < 4; ++b) {
    rcs_conv3x3<4, NP, NT, NW, F16>(acc, pk, ch, Q, lds_w, lds_w_addr, wave, lane);
    RC3_STAMP(3)
    rc3_epilogue<false, NP>(y, acc, x, lds_epi + (1 + 2 * b) * 192, h);
    rcs_pack<NP, NT, F16>(pk, y, amax);
    RC3_STAMP(2)
    rcs_conv3x3<4, NP, NT, NW, F16>(acc, pk, ch, Q, lds_w, lds_w_addr, wave, lane);
    RC3_STAMP(3)
    rc3_epilogue<true, NP>(x, acc, x, lds_epi + (2 + 2 * b) * 192, h);
    rcs_pack<NP, NT, F16>(pk, x, amax);
    RC3_STAMP(2)
  }
  if constexpr (F16) {
    if (!rcs_pk_f16_finite(amax)) atomicOr(Q.range_flag, 1u); /* (never in range: no lane enters) */
  }
  if (NT != 2) {
    /* the head weights were requested behind the last group */
    CO_WAIT_VMCNT(0);
    co_wg_barrier();
  }
  /* heads: the two 1x1 convolutions as one more split-precision step on the operands packed
   * after the last block (no tap shift); output rows 0..3 policy planes (h 0), 4..5 value (h 1) */
  float *feat_w = lds_feat + wave * (2 * NP) * 96; /* this wave's positions, workgroup order */
  u32x4 hw[NT][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t) hw[t][s] = *reinterpret_cast<const u32x4 *>(lds_head + ((s * NT + t) * 64 + lane) * 4);
#pragma unroll
  for (int np = 0; np < NP; ++np) {
    f32x16 h1;
#pragma unroll
    for (int i = 0; i < 16; ++i) h1[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 B[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int m = 0; m < 4; ++m) B[t][m] = pk[t][np][s][m];
      }
#pragma unroll
      for (int sum = 0; sum < NT; ++sum)
#pragma unroll
        for (int i = 0; i <= sum; ++i) h1 = rcs_mfma<F16>(hw[i][s], B[sum - i], h1);
    }
    const float4 b4 = *reinterpret_cast<const float4 *>(P.head_epi + 4 * h);
    const float4 a4 = *reinterpret_cast<const float4 *>(P.head_epi + 16 + 4 * h);
    const float4 c4 = *reinterpret_cast<const float4 *>(P.head_epi + 32 + 4 * h);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
    const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
    const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
    const int pw = np * 2 + p2; /* position of this lane within the wave */
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = h1[r] + bb[r];
      v = aa[r] * v + cc[r];
      v = v > 0.0f ? v : 0.0f;
      if (h == 0) feat_w[pw * 96 + c * 4 + r] = v;
      if (h == 1 && r < 2) feat_w[pw * 96 + 64 + c * 2 + r] = v;
    }
  }
  RC3_STAMP(4)
  __syncthreads();
  /* 16 NP positions = NP column tiles: waves 0 (, 1) run their policy heads, waves 2 (, 3) their
   * value heads */
  const float *lds_dense = reinterpret_cast<const float *>(lds_head + RCS_FRAG1_WORDS(NT));
  constexpr int wgpos = 2 * NP * NW, ntiles = (wgpos + 15) / 16;
  constexpr int ncols = wgpos < 16 ? wgpos : 16; /* the thin kernel's workgroup is half a column tile */
  if (wave < ntiles)
    rc_dense_policy(P, lds_dense, lds_feat + wave * 16 * 96, rows, row0 + wave * 16, lane, ncols);
  else if (wave >= 2 && wave < 2 + ntiles)
    rc_dense_value(P, lds_dense + 6144, lds_dense + 6144 + 2048, lds_feat + (wave - 2) * 16 * 96, rows,
                   row0 + (wave - 2) * 16, lane, ncols);
  RC3_STAMP(5)
#ifdef CO_PROF
  if (tid == 0) {
    atomicAdd(&rc3_prof[6], __builtin_readcyclecounter() - start_);
    atomicAdd(&rc3_prof[7], 1ull);
    atomicAdd(&rc3_prof[8], __builtin_amdgcn_s_memrealtime() - real_);
  }
#endif
}

/* NT = 2  <2>: throughput kernel, batches of more than RC3_SMALL_ROWS rows, 32 positions per workgroup;
 *         <1>: small-batch kernel, up to RC3_SMALL_ROWS rows, 16 positions per workgroup, one round
 * NT = 3  <1> only */
__global__ __launch_bounds__(512, 2) void co_k_rescnn_forward_x3(Rc3Params Q) { rcs_forward<2, 2>(Q); }
__global__ __launch_bounds__(512, 2) void co_k_rescnn_forward_x3_small(Rc3Params Q) { rcs_forward<1, 2>(Q); }
/* "f16x3": the two-term kernels with fp16 terms instead of bf16 ones -- x = fp16(x) + fp16(x - fp16(x)) keeps 22
 * significand bits per operand (bf16x3: 16), the three products w0 x0 + w0 x1 + w1 x0 drop terms of 2^-22 |w x|:
 * float32-class arithmetic at the MFMA cost of bf16x3.  fp16's exponent range is narrower (normal from 6.1e-5,
 * subnormal quantum 6e-8): remainders of small values lose relative, not absolute, accuracy -- measured against
 * float64 in tests/test_net_precision.py. */
/* (the kernels themselves: behind co_k_rescnn_forward_x6) */
/* (Capping this kernel at 168 registers so that a wave of the search kernel fits beside two of its waves on a SIMD was
 * measured: the network kernel alone 5 % slower, the generation 4 % slower -- the kernel trace shows 81 % of the search
 * kernel's time overlapping the other pool's network launches already, tools/overlap.py.) */
/* Thin batches (up to RC6_THIN_ROWS rows = 256 workgroups): four waves, one per SIMD, 8 positions per workgroup.  A batch
 * that does not fill the chip is as slow as ONE workgroup's pass over the 27 weight groups; with the MFMA pipe of a SIMD
 * to itself a wave finishes its 3456 MFMAs in half the time (the DPP operand shifts fit in their shadow).  The tail of a
 * generation, the arena and the analysis mode run such batches every iteration.  Same launch, same workgroups: the row
 * count on the device picks the path, and waves 4..7 of a thin workgroup leave at once (a second kernel that merely
 * returns would still queue 256 workgroups of 151 KB LDS behind the other pool's network launch). */
__global__ __launch_bounds__(512, 2) void co_k_rescnn_forward_x6(Rc3Params Q) {
  if (*Q.base.d_rows <= RC6_THIN_ROWS) {
    if (threadIdx.x >= 256) return;
    rcs_forward<1, 3, 4>(Q);
  } else {
    rcs_forward<1, 3, 8>(Q);
  }
}

/* The f16x3 kernels (see above rcs_forward): throughput kernel, 32 positions per workgroup; _small: batches up to
 * RC3_SMALL_ROWS rows, 16 positions per workgroup, and up to RC6_THIN_ROWS rows on the four-wave thin path (one wave per
 * SIMD, 8 positions per workgroup, waves 4..7 leave at once; see co_k_rescnn_forward_x6). */
#ifndef CO_RESCNN_PIXMAJOR_DEFAULT
#define CO_RESCNN_PIXMAJOR_DEFAULT 1 /* 0: a diagnostic build whose batches beyond RC3_SMALL_ROWS take the (position, pixel)-column kernel (tools/exp/pixmajor_ab.py) */
#endif
#if !CO_RESCNN_PIXMAJOR_DEFAULT
__global__ __launch_bounds__(512, 2) void co_k_rescnn_forward_h3(Rc3Params Q) { rcs_forward<2, 2, 8, true>(Q); }
#endif
__global__ __launch_bounds__(512, 2) void co_k_rescnn_forward_h3_small(Rc3Params Q) {
  const int rows = *Q.base.d_rows, rbase = rcp_small_begin(Q, rows);
  if (rows - rbase <= 0) return; /* the whole batch is the throughput kernel's */
  if (rows - rbase <= RC6_THIN_ROWS) {
    if (threadIdx.x >= 256) return;
    rcs_forward<1, 2, 4, true>(Q, rbase);
  } else {
    rcs_forward<1, 2, 8, true>(Q, rbase);
  }
}
#define RCH_LDS_WORDS(NP) RCS_LDS_WORDS(2, NP)
#define RCH_THREADS 512

/* ======================================================================
 * K6p: the f16x3 throughput kernel in PIXEL-MAJOR form (round 4).
 *
 * rcs_forward makes an MFMA column a (position, pixel) pair: a tap is a DPP shift of the activation registers, and the
 * 44 of 144 (pixel, tap) pairs that fall outside the 4x4 board multiply zeros -- 31 % of the matrix work of every 3x3
 * convolution.  Here a column is a POSITION (32 per workgroup) and every output pixel has its own accumulators,
 *     D_p[co, pos] += W_tap[co, ci] . X_q[ci, pos]        for the taps whose source pixel q = p + tap lies on the board,
 * so only the 100 real pairs are multiplied: 300 MFMAs per wave and convolution on average instead of 432.  The price:
 * the neighbour pixel's activations are another wave's registers, so activations travel through LDS -- 16 pixels x 4 K
 * steps x 2 fp16 terms x 1 KiB = 128 KB per workgroup, written by the epilogue of every convolution and read back as B
 * fragments -- and weights stream one tap (16 KB) at a time through the remaining 32 KB (a barrier per tap).
 * Waves 0-3 own an interior pixel (9 taps) and a corner (4), waves 4-7 two edge pixels of one side (6 + 6).
 *
 * Results are BIT-IDENTICAL to rcs_forward<.., 2, .., true>: the same weight fragments (same k-slot order), the same
 * products in the same order per accumulator (tap ascending, K step ascending, w0 x0, w0 x1, w1 x0), the same epilogue
 * expressions; the taps rcs_forward multiplies with zero padding add exact zeros there and are skipped here.  So a row
 * is evaluated to the same bits whichever kernel its batch size selects (SURVEY 8e invariant;
 * tests/test_net_precision.py::test_bf16x6_rows_do_not_depend_on_their_batch compares them). */
#define RCP_X_WORDS (16 * 4 * 2 * 256)   /* pixel x K step x term fragments: 128 KB */
#define RCP_TAP_WORDS RCS_CONV_CHUNK(2)   /* 16 KB */
#define RCP_LDS_WORDS (RCP_X_WORDS + 2 * RCP_TAP_WORDS)
#define RCP_NUM_TAPS 82                   /* 9 stem taps, 72 trunk taps, the 1x1 head fragments */

/* stream item g into weight buffer g & 1: a stem tap (4 KB), a trunk tap (16 KB) or the head fragments (8 KB) */
__device__ __forceinline__ void rcp_stage(const Rc3Params &Q, uint32_t lds_w_addr, int g, int wave, int lane) {
  const uint32_t dst = lds_w_addr + (uint32_t)(g & 1) * (RCP_TAP_WORDS * 4u);
  if (g < 9) rcs_stage_words(Q.wtrunk + g * RCS_STEM_CHUNK(2), dst, RCS_STEM_CHUNK(2), wave, lane);
  else if (g < 81) rcs_stage_words(Q.wtrunk + 9 * RCS_STEM_CHUNK(2) + (g - 9) * RCS_CONV_CHUNK(2), dst, RCS_CONV_CHUNK(2), wave, lane);
  else rcs_stage_words(Q.whead3, dst, RCS_FRAG1_WORDS(2), wave, lane);
}

/* one 3x3 convolution of this wave's two output pixels.  CS = K steps (1: stem, whose inputs are exact in fp16 -- only
 * their first term exists; 4: trunk) */
#ifdef CO_PROF
/* (sums in registers, written once at the end of the kernel: a stamp that touches memory would itself be waited for
 * by the loop's vmcnt wait) */
__device__ unsigned long long rcp_acc_dummy;
#define RCP_STAMP(slot)                                          \
  {                                                              \
    unsigned long long now_ = __builtin_readcyclecounter();      \
    pa[slot - 9] += now_ - tstamp;                               \
    tstamp = now_;                                               \
  }
#define RCP_TRACE(i) \
  if (trace_on) tr[i] = (unsigned)__builtin_readcyclecounter();
#define RCP_PROF_ARG , unsigned long long (&pa)[3], unsigned (&tr)[20], bool trace_on
#define RCP_PROF_PASS , pa, tr, false
#define RCP_PROF_PASS_TRACED , pa, tr, b == 1
#else
#define RCP_STAMP(slot)
#define RCP_TRACE(i)
#define RCP_PROF_ARG
#define RCP_PROF_PASS
#define RCP_PROF_PASS_TRACED
#endif
template <int CS>
__device__ __forceinline__ void rcp_conv3x3(f32x16 (&acc)[2][2], int &g, const Rc3Params &Q, const uint32_t *X, const uint32_t *Wb,
                                            uint32_t lds_w_addr, int P0, int P1, int valid0, int valid1, int wave, int lane RCP_PROF_ARG) {
#ifdef CO_PROF
  unsigned long long tstamp = __builtin_readcyclecounter();
#endif
  constexpr int XT = CS == 1 ? 1 : 2;
#pragma unroll
  for (int pi = 0; pi < 2; ++pi)
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[pi][to][i] = 0.0f;
  for (int tap = 0; tap < 9; ++tap, ++g) {
    RCP_STAMP(11)
    RCP_TRACE(2 * tap)
    CO_WAIT_VMCNT(0); /* this wave's pieces of item g have landed (requested one item ago) */
    RCP_STAMP(9)
    co_wg_barrier();  /* ... every wave's; everyone has left the other buffer and, at tap 0, has written its activations */
    RCP_STAMP(10)
    RCP_TRACE(2 * tap + 1)
    rcp_stage(Q, lds_w_addr, g + 1, wave, lane); /* (behind the first K step's MFMAs instead: 10 % slower, measured) */
    const bool v0 = (valid0 >> tap) & 1, v1 = (valid1 >> tap) & 1;
    if (!v0 && !v1) continue;
    const uint32_t *wb = Wb + (g & 1) * RCP_TAP_WORDS + lane * 4;
    const int dq = (tap / 3 - 1) * 4 + (tap % 3 - 1);
    const uint32_t *x0 = X + ((P0 + dq) * 4 * 2) * 256 + lane * 4, *x1 = X + ((P1 + dq) * 4 * 2) * 256 + lane * 4;
#pragma unroll
    for (int s = 0; s < CS; ++s) {
      u32x4 a[2][2];
#pragma unroll
      for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t][to] = *reinterpret_cast<const u32x4 *>(wb + ((s * 2 + to) * 2 + t) * 256);
      if (v0) {
        u32x4 b[XT];
#pragma unroll
        for (int t = 0; t < XT; ++t) b[t] = *reinterpret_cast<const u32x4 *>(x0 + (s * 2 + t) * 256);
#pragma unroll
        for (int sum = 0; sum < 2; ++sum)
#pragma unroll
          for (int i = 0; i <= sum; ++i)
            if (sum - i < XT) {
#pragma unroll
              for (int to = 0; to < 2; ++to) acc[0][to] = rcs_mfma<true>(a[i][to], b[sum - i], acc[0][to]);
            }
      }
      if (v1) {
        u32x4 b[XT];
#pragma unroll
        for (int t = 0; t < XT; ++t) b[t] = *reinterpret_cast<const u32x4 *>(x1 + (s * 2 + t) * 256);
#pragma unroll
        for (int sum = 0; sum < 2; ++sum)
#pragma unroll
          for (int i = 0; i <= sum; ++i)
            if (sum - i < XT) {
#pragma unroll
              for (int to = 0; to < 2; ++to) acc[1][to] = rcs_mfma<true>(a[i][to], b[sum - i], acc[1][to]);
            }
      }
    }
  }
  RCP_STAMP(11)
  RCP_TRACE(18)
}

/* conv bias -> BatchNorm affine (-> + skip) -> ReLU (rc3_epilogue's expressions), then the two fp16 terms of the result
 * go to LDS as the B fragments of the next convolution.  KEEP: the fp32 result replaces `x` (the skip of the block).
 * Everything that does not touch LDS -- the constants' loads, the arithmetic, the split -- runs BEFORE the barrier that
 * waits for the other waves to finish reading the old activations: a wave that is done with its taps works on its
 * epilogue while the slower SIMDs still multiply, and only the sixteen stores per pixel stand behind the barrier. */
template <bool ADD_SKIP, bool KEEP>
__device__ __forceinline__ void rcp_epilogue(float (&x)[2][2][16], const f32x16 (&acc)[2][2], const float *epi, uint32_t *X, int P0, int P1,
                                             int h, int lane, uint32_t &amax) {
  u32x4 hi[2][2][2], lo[2][2][2]; /* [pixel][tile][half of the tile's registers] */
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    float out[2][16];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      const int chn = 32 * T + 8 * gg + 4 * h;
      const float4 b4 = *reinterpret_cast<const float4 *>(epi + chn);
      const float4 a4 = *reinterpret_cast<const float4 *>(epi + 64 + chn);
      const float4 c4 = *reinterpret_cast<const float4 *>(epi + 128 + chn);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
      const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
      const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float cb = __builtin_fmaf(aa[i], bb[i], cc[i]);
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
          float v = __builtin_fmaf(aa[i], acc[pi][T][4 * gg + i], cb);
          if (ADD_SKIP) v = x[pi][T][4 * gg + i] + v;
          v = v > 0.0f ? v : 0.0f;
          out[pi][4 * gg + i] = v;
        }
      }
    }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float u0 = out[pi][8 * a2 + 2 * m], u1 = out[pi][8 * a2 + 2 * m + 1];
          uint32_t t[2];
          rcs_split<2, true>(u0, u1, t);
          amax = rcs_pk_max_f16(amax, t[0]);
          hi[pi][T][a2][m] = t[0];
          lo[pi][T][a2][m] = t[1];
        }
      }
      if (KEEP) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[pi][T][i] = out[pi][i];
      }
    }
  }
  co_wg_barrier(); /* every wave has read the activations this convolution consumed: they may be overwritten */
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {
    const int p = pi ? P1 : P0;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2) {
        const int sidx = 2 * T + a2;
        *reinterpret_cast<u32x4 *>(X + ((p * 4 + sidx) * 2 + 0) * 256 + lane * 4) = hi[pi][T][a2];
        *reinterpret_cast<u32x4 *>(X + ((p * 4 + sidx) * 2 + 1) * 256 + lane * 4) = lo[pi][T][a2];
      }
  }
}

__global__ __launch_bounds__(512, 2) void co_k_rescnn_forward_h3p(Rc3Params Q) {
  const RcParams &P = Q.base;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const int rows = *P.d_rows;
  const int row0 = blockIdx.x * 32;
  if (row0 >= rcp_small_begin(Q, rows)) return; /* beyond the batch, or in the share of the small-batch kernel */
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, h = lane >> 5, n = lane & 31;
#ifdef CO_PROF
  unsigned long long stamp_ = __builtin_readcyclecounter();
  const unsigned long long start_ = stamp_, real_ = __builtin_amdgcn_s_memrealtime();
  unsigned long long pa[3] = {0ull, 0ull, 0ull}, ph[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  unsigned tr[20] = {};
#define RCP_PHASE(slot)                                      \
  {                                                          \
    unsigned long long now_ = __builtin_readcyclecounter();  \
    ph[slot] += now_ - stamp_;                               \
    stamp_ = now_;                                           \
  }
#else
#define RCP_PHASE(slot)
#endif
  uint32_t *X = lds_dyn;
  const uint32_t *Wb = lds_dyn + RCP_X_WORDS;
  const uint32_t lds_w_addr = co_lds_addr(lds_dyn) + RCP_X_WORDS * 4u;
  /* this wave's two output pixels and their taps on the board */
  /* Waves w and w + 4 share a SIMD (a workgroup's waves go round the four SIMDs), and every tap ends at a barrier: what
   * a tap costs is the SIMD with the most (pixel, tap) pairs on the board at THAT tap.  The four pixels of a SIMD are
   * chosen so that every tap is spread evenly -- {0, 5, 6, 15}, {1, 4, 7, 13}, {2, 8, 11, 14}, {3, 9, 10, 12}: at most
   * 4, 3 or 3 pairs per SIMD at the centre, side and diagonal taps, 28 slots per convolution where the board has 25 per
   * SIMD on average (the optimum over all 2.6 M partitions; an interior + a corner and two neighbouring edge pixels per
   * wave: 34; the (position, pixel)-column kernel multiplies all 36). */
  const int P0 = wave == 0 ? 5 : wave == 1 ? 1 : wave == 2 ? 2 : wave == 3 ? 9 : wave == 4 ? 6 : wave == 5 ? 4 : wave == 6 ? 8 : 10;
  const int P1 = wave == 0 ? 15 : wave == 1 ? 13 : wave == 2 ? 14 : wave == 3 ? 12 : wave == 4 ? 0 : wave == 5 ? 7 : wave == 6 ? 11 : 3;
  int valid0 = 0, valid1 = 0;
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    if ((P0 >> 2) + dy >= 0 && (P0 >> 2) + dy < 4 && (P0 & 3) + dx >= 0 && (P0 & 3) + dx < 4) valid0 |= 1 << tap;
    if ((P1 >> 2) + dy >= 0 && (P1 >> 2) + dy < 4 && (P1 & 3) + dx >= 0 && (P1 & 3) + dx < 4) valid1 |= 1 << tap;
  }
  /* input planes of this wave's pixels as the stem's B fragments (K step 0, first term: board bits and k / 4 are exact in
   * fp16).  k-slot (h, j) <-> channel 8 (j / 4) + 4 h + j % 4: h 0 = the cell's four board bits, reserves 4..5 and padding;
   * h 1 = reserves 0..3 and zeros (rcs_forward's planes) */
  {
    const int pos = row0 + n;
    float4 vq[2], v1 = make_float4(0.f, 0.f, 0.f, 0.f);
    vq[0] = vq[1] = v1;
    if (pos < rows) {
      const float *row = P.in + rc_in_row(P, pos) * CO_STATE_STRIDE;
      vq[0] = *reinterpret_cast<const float4 *>(row + (h == 0 ? 4 * P0 : 64));
      vq[1] = *reinterpret_cast<const float4 *>(row + (h == 0 ? 4 * P1 : 64));
      v1 = *reinterpret_cast<const float4 *>(row + (h == 0 ? 68 : 72));
    }
    rcp_stage(Q, lds_w_addr, 0, wave, lane); /* behind the input loads: vmcnt retires in issue order */
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int p = pi ? P1 : P0;
      uint32_t t[1];
      u32x4 f;
      rcs_split<1, true>(vq[pi].x, vq[pi].y, t);
      f[0] = t[0];
      rcs_split<1, true>(vq[pi].z, vq[pi].w, t);
      f[1] = t[0];
      rcs_split<1, true>(v1.x, v1.y, t);
      f[2] = t[0];
      rcs_split<1, true>(v1.z, v1.w, t);
      f[3] = t[0];
      *reinterpret_cast<u32x4 *>(X + ((p * 4 + 0) * 2 + 0) * 256 + lane * 4) = f;
    }
  }
  uint32_t amax = 0u; /* (packed fp16 pair: rcs_pk_max_f16) */
  f32x16 acc[2][2];
  float x[2][2][16];
#pragma unroll
  for (int pi = 0; pi < 2; ++pi)
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int i = 0; i < 16; ++i) x[pi][T][i] = 0.0f;
  const float *epi = P.epi; /* (global: the 160 KB of LDS hold activations and weights) */
  int g = 0;
  RCP_PHASE(0)
  rcp_conv3x3<1>(acc, g, Q, X, Wb, lds_w_addr, P0, P1, valid0, valid1, wave, lane RCP_PROF_PASS);
  RCP_PHASE(1)
  rcp_epilogue<false, true>(x, acc, epi, X, P0, P1, h, lane, amax);
  RCP_PHASE(2)
  for (int b = 0; b < 4; ++b) {
    rcp_conv3x3<4>(acc, g, Q, X, Wb, lds_w_addr, P0, P1, valid0, valid1, wave, lane RCP_PROF_PASS);
    RCP_PHASE(3)
    rcp_epilogue<false, false>(x, acc, epi + (1 + 2 * b) * 192, X, P0, P1, h, lane, amax);
    RCP_PHASE(2)
    rcp_conv3x3<4>(acc, g, Q, X, Wb, lds_w_addr, P0, P1, valid0, valid1, wave, lane RCP_PROF_PASS_TRACED);
    RCP_PHASE(3)
    rcp_epilogue<true, true>(x, acc, epi + (2 + 2 * b) * 192, X, P0, P1, h, lane, amax);
    RCP_PHASE(2)
  }
  if (!rcs_pk_f16_finite(amax)) atomicOr(Q.range_flag, 1u); /* (never in range: no lane enters) */
  /* heads: item 81 = the 1x1 convolutions' fragments (rows 0..3 policy planes, 4..5 value planes), in buffer 1; the head
   * features of the 32 positions go to buffer 0, which tap 80 has left */
  CO_WAIT_VMCNT(0);
  co_wg_barrier();
  {
    const uint32_t *hwb = Wb + (81 & 1) * RCP_TAP_WORDS + lane * 4;
    float *feat = reinterpret_cast<float *>(lds_dyn + RCP_X_WORDS); /* buffer 0: [32 positions][96] */
    u32x4 hw[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) hw[t][s] = *reinterpret_cast<const u32x4 *>(hwb + (s * 2 + t) * 256);
    const float4 b4 = *reinterpret_cast<const float4 *>(P.head_epi + 4 * h);
    const float4 a4 = *reinterpret_cast<const float4 *>(P.head_epi + 16 + 4 * h);
    const float4 c4 = *reinterpret_cast<const float4 *>(P.head_epi + 32 + 4 * h);
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
    const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
    const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int p = pi ? P1 : P0;
      f32x16 h1;
#pragma unroll
      for (int i = 0; i < 16; ++i) h1[i] = 0.0f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4 b0 = *reinterpret_cast<const u32x4 *>(X + ((p * 4 + s) * 2 + 0) * 256 + lane * 4);
        const u32x4 b1 = *reinterpret_cast<const u32x4 *>(X + ((p * 4 + s) * 2 + 1) * 256 + lane * 4);
        h1 = rcs_mfma<true>(hw[0][s], b0, h1);
        h1 = rcs_mfma<true>(hw[0][s], b1, h1);
        h1 = rcs_mfma<true>(hw[1][s], b0, h1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = h1[r] + bb[r];
        v = aa[r] * v + cc[r];
        v = v > 0.0f ? v : 0.0f;
        if (h == 0) feat[n * 96 + p * 4 + r] = v;
        if (h == 1 && r < 2) feat[n * 96 + 64 + p * 2 + r] = v;
      }
    }
    __syncthreads();
    /* 32 positions = two column tiles: waves 0, 1 their policy heads, waves 2, 3 their value heads; the dense weights
     * straight from global memory (fp32, MFMA order, shared by every workgroup: L2) */
    RCP_PHASE(4)
    if (wave < 2) rc_dense_policy(P, P.wpol, feat + wave * 16 * 96, rows, row0 + wave * 16, lane, 16);
    else if (wave < 4) rc_dense_value(P, P.wv1, P.wv2, feat + (wave - 2) * 16 * 96, rows, row0 + (wave - 2) * 16, lane, 16);
    RCP_PHASE(5)
#ifdef CO_PROF
    if (tid == 0) {
      atomicAdd(&rc3_prof[6], __builtin_readcyclecounter() - start_);
      atomicAdd(&rc3_prof[7], 1ull);
      atomicAdd(&rc3_prof[8], __builtin_amdgcn_s_memrealtime() - real_);
      for (int i = 0; i < 6; ++i) atomicAdd(&rc3_prof[i], ph[i]);
      atomicAdd(&rc3_prof[9], pa[0]);
      atomicAdd(&rc3_prof[10], pa[1]);
      atomicAdd(&rc3_prof[11], pa[2]);
    }
    if (blockIdx.x == 0 && lane == 0)
      for (int i = 0; i < 20; ++i) rc3_trace[wave * 20 + i] = tr[i];
#endif
  }
}

/* ------------------------------------------------------------------ host */
struct ResCnnNet : CoNet {
  std::vector<float *> bufs;
  RcParams P;
  size_t cap;

  float *upload(const std::vector<float> &h, rt_stream_t s) {
    float *d = nullptr;
    rt_malloc((void **)&d, h.size() * 4, s);
    rt_h2d(d, h.data(), h.size() * 4, s);
    bufs.push_back(d);
    return d;
  }

  ResCnnNet(const float *w, size_t max_rows, rt_stream_t s) : cap(max_rows) {
    const float *p = w;
    std::vector<float> trunk(RC_TRUNK_FLOATS, 0.0f), epi((size_t)RC_NUM_CONVS * 192, 0.0f);
    auto bn_fold = [](const float *ga, const float *be, const float *mu, const float *va, int n, float *a, float *c) {
      for (int i = 0; i < n; ++i) {
        a[i] = (float)((double)ga[i] / sqrt((double)va[i] + CO_BN_EPS));
        c[i] = (float)((double)be[i] - (double)mu[i] * (double)a[i]);
      }
    };
    size_t off = 0;
    for (int cv = 0; cv < RC_NUM_CONVS; ++cv) {
      const int cin = cv == 0 ? 10 : 64;
      const int ct = cv == 0 ? 1 : 4;
      const size_t chunk = cv == 0 ? RC_STEM_CHUNK : RC_CONV_CHUNK;
      const float *K = p; /* [3][3][cin][64] */
      for (int tap = 0; tap < 9; ++tap)
        for (int t = 0; t < ct; ++t)
          for (int r = 0; r < 4; ++r)
            for (int to = 0; to < 4; ++to)
              for (int q = 0; q < 4; ++q)
                for (int i = 0; i < 16; ++i) {
                  int ci = 16 * t + 4 * q + r, co = 16 * to + i;
                  float v = ci < cin ? K[((size_t)tap * cin + ci) * 64 + co] : 0.0f;
                  trunk[off + (size_t)tap * chunk + ((size_t)(t * 4 + r) * 4 + to) * 64 + 16 * q + i] = v;
                }
      off += 9 * chunk;
      p += (size_t)9 * cin * 64;
      const float *b = p, *ga = b + 64, *be = ga + 64, *mu = be + 64, *va = mu + 64;
      for (int i = 0; i < 64; ++i) epi[(size_t)cv * 192 + i] = b[i];
      bn_fold(ga, be, mu, va, 64, &epi[(size_t)cv * 192 + 64], &epi[(size_t)cv * 192 + 128]);
      p = va + 64;
    }
    /* policy head */
    const float *pk = p, *pb = pk + 64 * 4, *pga = pb + 4, *pbe = pga + 4, *pmu = pbe + 4, *pva = pmu + 4;
    const float *pdk = pva + 4, *pdb = pdk + 64 * 96;
    const float *vk = pdb + 96, *vb = vk + 64 * 2, *vga = vb + 2, *vbe = vga + 2, *vmu = vbe + 2, *vva = vmu + 2;
    const float *vd1k = vva + 2, *vd1b = vd1k + 32 * 64, *vd2k = vd1b + 64, *vd2b = vd2k + 64;
    std::vector<float> whead(16 * 64, 0.0f), hepi(48, 0.0f), wpol(16 * 6 * 64, 0.0f), bpol(pdb, pdb + 96);
    std::vector<float> wv1(8 * 4 * 64, 0.0f), bv1(vd1b, vd1b + 64), wv2(16 * 64, 0.0f), bv2(vd2b, vd2b + 1);
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 4; ++r)
        for (int q = 0; q < 4; ++q)
          for (int i = 0; i < 16; ++i) {
            int k = 16 * t + 4 * q + r;
            float v = i < 4 ? pk[k * 4 + i] : i < 6 ? vk[k * 2 + (i - 4)] : 0.0f;
            whead[(size_t)(t * 4 + r) * 64 + 16 * q + i] = v;
            wv2[(size_t)(t * 4 + r) * 64 + 16 * q + i] = i == 0 ? vd2k[k] : 0.0f;
          }
    for (int i = 0; i < 4; ++i) hepi[i] = pb[i];
    for (int i = 0; i < 2; ++i) hepi[4 + i] = vb[i];
    bn_fold(pga, pbe, pmu, pva, 4, &hepi[16], &hepi[32]);
    bn_fold(vga, vbe, vmu, vva, 2, &hepi[16 + 4], &hepi[32 + 4]);
    for (int s2 = 0; s2 < 16; ++s2)
      for (int to = 0; to < 6; ++to)
        for (int q = 0; q < 4; ++q)
          for (int i = 0; i < 16; ++i) wpol[((size_t)s2 * 6 + to) * 64 + 16 * q + i] = pdk[(size_t)(4 * s2 + q) * 96 + 16 * to + i];
    for (int s2 = 0; s2 < 8; ++s2)
      for (int to = 0; to < 4; ++to)
        for (int q = 0; q < 4; ++q)
          for (int i = 0; i < 16; ++i) wv1[((size_t)s2 * 4 + to) * 64 + 16 * q + i] = vd1k[(size_t)(4 * s2 + q) * 64 + 16 * to + i];
    memset(&P, 0, sizeof P);
    P.wtrunk = upload(trunk, s);
    P.epi = upload(epi, s);
    P.whead = upload(whead, s);
    P.head_epi = upload(hepi, s);
    P.wpol = upload(wpol, s);
    P.bpol = upload(bpol, s);
    P.wv1 = upload(wv1, s);
    P.bv1 = upload(bv1, s);
    P.wv2 = upload(wv2, s);
    P.bv2 = upload(bv2, s);
    rt_sync(s);
  }
  ~ResCnnNet() override {
    for (float *d : bufs) rt_free(d);
  }
  size_t max_rows() const override { return cap; }
  int kind() const override { return CO_NET_RESCNN4; }
  double flop_per_row() const override {
    return 2.0 * 16 * 9 * (10 * 64 + 8 * 64 * 64) + 2.0 * 16 * 64 * 6 + 2.0 * 64 * 96 + 2.0 * 32 * 64 + 2.0 * 64;
  }
  void forward(const float *d_in, int32_t rows_cap, const int32_t *d_rows, float *d_eval, float *d_probs,
               rt_stream_t s, const CoNetIO &io = CoNetIO()) override {
    int grid = (rows_cap + RC_POS_PER_WG - 1) / RC_POS_PER_WG;
    if (grid < 1) return;
    RcParams p = P;
    p.io = io;
    p.in = d_in;
    p.d_rows = d_rows;
    p.eval = d_eval;
    p.probs = d_probs;
    hipLaunchKernelGGL(co_k_rescnn_forward, dim3(grid), dim3(256), 0, s, p);
    RT_CHECK(hipGetLastError());
  }
};

static inline uint16_t rc_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float rc_bf16_to_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
/* float -> IEEE binary16, round to nearest even, subnormals kept (what v_cvt_f16_f32 gives) */
static inline uint16_t rc_f16_rne(float f) {
  _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static inline float rc_f16_to_f(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}
/* v -> nt 16-bit terms: bf16(v) (f16: fp16(v)), the same of the remainder, ... (the device's rcs_split) */
static inline void rc_bf16_terms(float v, int nt, uint16_t *t, bool f16 = false) {
  if (f16 && !(fabsf(v) <= CO_F16_MAX))
    throw std::invalid_argument("rescnn4h3: a convolution weight is " + std::to_string(v) +
                                ", beyond the fp16 range of the f16x3 kernels: use rescnn4x6");
  for (int i = 0; i < nt; ++i) {
    t[i] = f16 ? rc_f16_rne(v) : rc_bf16_rne(v);
    v = v - (f16 ? rc_f16_to_f(t[i]) : rc_bf16_to_f(t[i]));
  }
}

/* the split-precision kernels: NT = 2 (bf16x3) or 3 (bf16x6, float32-equivalent) */
struct ResCnnSplitNet : ResCnnNet {
  int nt;
  bool f16;
  bool pixmajor = false; /* f16: batches beyond RC3_SMALL_ROWS on the pixel-major kernel (K6p) */
  int num_cus = 256;     /* (a pass of that kernel = one workgroup per CU) */
  uint32_t *d_trunk3 = nullptr;
  uint32_t *d_whead3 = nullptr;
  uint32_t *d_epi3 = nullptr;
  uint32_t *d_range = nullptr; /* f16: the kernels' out-of-range flag */
  ResCnnSplitNet(const float *w, size_t max_rows, rt_stream_t s, int nterms, bool fp16 = false)
      : ResCnnNet(w, max_rows, s), nt(nterms), f16(fp16) {
    const size_t stem_chunk = (size_t)512 * nt, conv_chunk = (size_t)2048 * nt;
    std::vector<uint32_t> tr(9 * stem_chunk + 72 * conv_chunk, 0u);
    const float *p = w;
    size_t off = 0;
    uint16_t tv[3];
    for (int cv = 0; cv < RC_NUM_CONVS; ++cv) {
      const int cin = cv == 0 ? 10 : 64;
      const int cs = cv == 0 ? 1 : 4;
      const size_t chunk = cv == 0 ? stem_chunk : conv_chunk;
      const float *K = p;
      for (int tap = 0; tap < 9; ++tap)
        for (int st = 0; st < cs; ++st)
          for (int to = 0; to < 2; ++to)
            for (int h = 0; h < 2; ++h)
              for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 8; ++j) {
                  /* step st = 2T + a; k-slot (h, j) <-> channel 32T + 4h + 8(2a + j/4) + j%4 */
                  int T = st >> 1, a = st & 1;
                  int ci = 32 * T + 4 * h + 8 * (2 * a + (j >> 2)) + (j & 3);
                  int co = 32 * to + i;
                  float v = ci < cin ? K[((size_t)tap * cin + ci) * 64 + co] : 0.0f;
                  rc_bf16_terms(v, nt, tv, f16);
                  size_t lane = 32 * h + i;
                  for (int t = 0; t < nt; ++t) {
                    size_t wd = off + (size_t)tap * chunk + ((((size_t)st * 2 + to) * nt + t) * 64 + lane) * 4 + j / 2;
                    tr[wd] |= (uint32_t)tv[t] << (16 * (j & 1));
                  }
                }
      off += 9 * chunk;
      p += (size_t)9 * cin * 64 + 5 * 64;
    }
    /* 1x1 head convolutions as term fragments of one more K loop (same k-slot order as the
     * trunk): output row i = 0..3 policy planes, 4..5 value planes, the rest zero */
    const float *pk = p, *vk = pk + 64 * 4 + 4 * 5 + 64 * 96 + 96;
    const size_t frag1 = (size_t)4 * nt * 256, head_words = frag1 + 6144 + 2048 + 1024;
    std::vector<uint32_t> wh3(frag1, 0u);
    for (int st = 0; st < 4; ++st)
      for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 8; ++j) {
            int T = st >> 1, a = st & 1;
            int k = 32 * T + 4 * h + 8 * (2 * a + (j >> 2)) + (j & 3);
            float v = i < 4 ? pk[k * 4 + i] : i < 6 ? vk[k * 2 + (i - 4)] : 0.0f;
            rc_bf16_terms(v, nt, tv, f16);
            size_t lane = 32 * h + i;
            for (int t = 0; t < nt; ++t)
              wh3[(((size_t)st * nt + t) * 64 + lane) * 4 + j / 2] |= (uint32_t)tv[t] << (16 * (j & 1));
          }
    rt_malloc((void **)&d_trunk3, tr.size() * 4, s);
    rt_h2d(d_trunk3, tr.data(), tr.size() * 4, s);
    rt_malloc((void **)&d_whead3, head_words * 4, s);
    rt_h2d(d_whead3, wh3.data(), wh3.size() * 4, s);
    rt_d2d(d_whead3 + frag1, P.wpol, 6144 * 4, s); /* the dense weights in the base class's MFMA order */
    rt_d2d(d_whead3 + frag1 + 6144, P.wv1, 2048 * 4, s);
    rt_d2d(d_whead3 + frag1 + 6144 + 2048, P.wv2, 1024 * 4, s);
    if (f16) rt_malloc((void **)&d_range, 4, s);
    rt_malloc((void **)&d_epi3, (size_t)RC3_EPI_WORDS * 4, s); /* zero-filled: the padding is staged too */
    rt_d2d(d_epi3, P.epi, (size_t)RC_NUM_CONVS * 192 * 4, s);
    if (nt == 2) {
      RT_CHECK(hipFuncSetAttribute((const void *)co_k_rescnn_forward_x3, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   RCS_LDS_WORDS(2, 2) * 4));
      RT_CHECK(hipFuncSetAttribute((const void *)co_k_rescnn_forward_x3_small, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   RCS_LDS_WORDS(2, 1) * 4));
#if !CO_RESCNN_PIXMAJOR_DEFAULT
      RT_CHECK(hipFuncSetAttribute((const void *)co_k_rescnn_forward_h3, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   RCH_LDS_WORDS(2) * 4));
#endif
      RT_CHECK(hipFuncSetAttribute((const void *)co_k_rescnn_forward_h3p, hipFuncAttributeMaxDynamicSharedMemorySize, RCP_LDS_WORDS * 4));
      {
        pixmajor = f16 && CO_RESCNN_PIXMAJOR_DEFAULT != 0;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
          num_cus = prop.multiProcessorCount;
      }
      RT_CHECK(hipFuncSetAttribute((const void *)co_k_rescnn_forward_h3_small, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   RCH_LDS_WORDS(1) * 4));
    } else {
      RT_CHECK(hipFuncSetAttribute((const void *)co_k_rescnn_forward_x6, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   RCS_LDS_WORDS(3, 1) * 4));

    }
    rt_sync(s);
  }
  ~ResCnnSplitNet() override {
    rt_free(d_trunk3);
    rt_free(d_whead3);
    rt_free(d_epi3);
    rt_free(d_range);
  }
  bool range_exceeded(rt_stream_t s) override {
    if (!d_range) return false;
    uint32_t flag = 0;
    rt_d2h(&flag, d_range, 4, s);
    rt_sync(s);
    return flag != 0;
  }
  int kind() const override { return f16 ? CO_NET_RESCNN4_H3 : nt == 2 ? CO_NET_RESCNN4_X3 : CO_NET_RESCNN4_X6; }
  void forward(const float *d_in, int32_t rows_cap, const int32_t *d_rows, float *d_eval, float *d_probs,
               rt_stream_t s, const CoNetIO &io = CoNetIO()) override {
    if (rows_cap < 1) return;
    Rc3Params q;
    q.base = P;
    q.base.io = io;
    q.base.in = d_in;
    q.base.d_rows = d_rows;
    q.base.eval = d_eval;
    q.base.probs = d_probs;
    q.wtrunk = d_trunk3;
    q.whead3 = d_whead3;
    q.epi3 = d_epi3;
    q.range_flag = d_range;
    q.pass_rows = pixmajor && io.alone ? 32 * num_cus : 0;
    if (nt == 2) {
      /* both kernels are queued; the row count on the device decides which one works (the other's
       * workgroups return at once).  Batches that can exceed RC3_SMALL_ROWS need the throughput kernel. */
      const int small_rows = rows_cap < RC3_SMALL_ROWS ? rows_cap : RC3_SMALL_ROWS;
      /* enough workgroups for either path of the f16 kernel: 16 positions each, or 8 on its thin path (<= RC6_THIN_ROWS rows) */
      const int thin_rows = rows_cap < RC6_THIN_ROWS ? rows_cap : RC6_THIN_ROWS;
      const int small_grid = f16 && (thin_rows + 7) / 8 > (small_rows + 15) / 16 ? (thin_rows + 7) / 8 : (small_rows + 15) / 16;
      hipLaunchKernelGGL(f16 ? co_k_rescnn_forward_h3_small : co_k_rescnn_forward_x3_small, dim3(small_grid), dim3(512),
                         (f16 ? RCH_LDS_WORDS(1) : RCS_LDS_WORDS(2, 1)) * 4, s, q);
      if (rows_cap > RC3_SMALL_ROWS) {
        if (pixmajor)
          hipLaunchKernelGGL(co_k_rescnn_forward_h3p, dim3((rows_cap + 31) / 32), dim3(512), RCP_LDS_WORDS * 4, s, q);
#if !CO_RESCNN_PIXMAJOR_DEFAULT
        else if (f16)
          hipLaunchKernelGGL(co_k_rescnn_forward_h3, dim3((rows_cap + 31) / 32), dim3(RCH_THREADS), RCH_LDS_WORDS(2) * 4, s, q);
#endif
        else
          hipLaunchKernelGGL(co_k_rescnn_forward_x3, dim3((rows_cap + 31) / 32), dim3(512), RCS_LDS_WORDS(2, 2) * 4, s, q);
      }
    } else {
      /* enough workgroups for either path: 16 positions each in the throughput path, 8 in the thin one (<= 2048 rows) */
      const int thin_rows = rows_cap < RC6_THIN_ROWS ? rows_cap : RC6_THIN_ROWS;
      const int grid = (rows_cap + 15) / 16 > (thin_rows + 7) / 8 ? (rows_cap + 15) / 16 : (thin_rows + 7) / 8;
      hipLaunchKernelGGL(co_k_rescnn_forward_x6, dim3(grid), dim3(512), RCS_LDS_WORDS(3, 1) * 4, s, q);
    }
    RT_CHECK(hipGetLastError());
  }
};

CoNet *co_rescnn_create(const float *weights, size_t n_floats, size_t max_rows, rt_stream_t s) {
  if (n_floats != (size_t)RC_NUM_WEIGHTS) return nullptr;
  return new ResCnnNet(weights, max_rows, s);
}

CoNet *co_rescnn_split_create(const float *weights, size_t n_floats, size_t max_rows, rt_stream_t s, int nterms, bool f16) {
  if (n_floats != (size_t)RC_NUM_WEIGHTS || (nterms != 2 && nterms != 3) || (f16 && nterms != 2)) return nullptr;
  return new ResCnnSplitNet(weights, max_rows, s, nterms, f16);
}
