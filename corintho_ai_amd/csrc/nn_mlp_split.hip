// nn_mlp_split.hip -- K5x3 / K5x6: the reference's policy/value network (12 x [Dense(100) -> ReLU ->
// BatchNorm] -> {Dense(1) tanh, Dense(96) softmax}, wrapper.py:256-271) as ONE fused gfx950
// kernel at split precision on v_mfma_f32_32x32x16_bf16.
//
// Same network, same flat weights and the same function as K5 (nn_mlp.hip, fp32 MFMA); what
// changes is the arithmetic of the thirteen matrix products.  Both operands are written as a sum
// of NT bf16 terms (the value rounded to bf16, then the successive remainders, each exact in
// float32) and the product is expanded into the terms w_i x_j with i + j <= NT - 1, accumulated in
// fp32 by the MFMA:
//   NT = 2 (CO_NET_MLP12X100_X3, "bf16x3"): 16 significand bits, 3 MFMAs, <= 1e-4 from float32;
//   NT = 3 (CO_NET_MLP12X100_X6, "bf16x6"): the three terms ARE the float32 value, 6 MFMAs, the
//          dropped terms are <= 2^-24 of a product: float32-equivalent (tests/test_net_precision.py
//          holds its error against a float64 restatement to that of K5's fp32 MFMA chain).
// The dense layers of this network are too small for fp32 MFMA to pay: K5 spends as long on them
// as the whole tree search takes.
//
// Mapping (the one of the residual CNN kernel, nn_rescnn.hip).  Transposed evaluation:
// out^T[feature][row] = W^T[feature][k] act^T[k][row]; a wave owns 32 batch rows = one MFMA
// column tile and all 128 (padded) output features = four 32-row tiles, 64 accumulator
// registers.  The accumulator layout (lane = (h, row), register 4g + i of tile T = feature
// 32T + 8g + 4h + i) is the B-operand layout of the next layer when its K steps are taken in the
// order  step s = 2T + a, k-slot (h, j) <-> feature 32T + 8(2a + j/4) + 4h + j%4,  so activations
// stay in registers through all 13 layers; the weights are pre-permuted (and pre-split into
// bf16 terms) on the host into that fragment order and stream through LDS by LDS-DMA.
// BatchNorm of layer l is folded into the weights and bias of layer l + 1 on the host (in
// float64, as the TFLite converter does for the reference's own checkpoints), the bias is the
// initial value of the accumulators, so a layer's epilogue is ReLU + the split.  tanh and the
// 96-way softmax are fused into the last layer.  Fixed k order, no batch-dependent tiling: a
// row's result does not depend on its batch.
//
// Weight stream: a layer is two CHUNKS of four K steps (K = 112 = 7 steps, the eighth is padding
// that is staged but never multiplied; the input layer has K = 80 = 5 steps), every chunk the same
// size (4 steps + a 4 KiB bias piece = 13 LDS-DMA instructions per wave), through a ring of three
// LDS slots: while chunk c computes, chunk c + 1 has landed or is landing and chunk c + 2 is
// requested -- the wait at the top of a chunk is `vmcnt(13)` (everything but the youngest chunk),
// not a drain.  (Round 1 staged whole layers through two slots and drained at every layer; the
// kernel waited for the stream about half of its time.)
//
// Geometry: 256 threads = 4 waves x 32 rows = 128 rows per workgroup, one workgroup per CU.
// The input layer's operands are exact in bf16 (0/1 and k/4): only their first term exists.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "engine_defs.h"
#include "lds_dma.h"
#include "nn.h"

typedef float m3_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 m3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 m3_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 m3_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 m3_f16x2 __attribute__((ext_vector_type(2)));
typedef float m3_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t m3_u32x4 __attribute__((ext_vector_type(4)));

#define M3_NLAYERS 13
#define M3_NCHUNKS (2 * M3_NLAYERS)
#define M3_STEPS 7      /* hidden layers and heads: K = 112 */
#define M3_STEPS_L0 5   /* input layer: K = 80 */
#define M3_STEP_WORDS(NT) (4 * (NT) * 64 * 4) /* 4 output tiles x NT terms x 64 lanes x 4 words */
#define M3_BIAS_WORDS 1024                     /* 128 biases in a 4 KiB piece: one LDS-DMA per wave */
#define M3_CHUNK_WORDS(NT) (4 * M3_STEP_WORDS(NT) + M3_BIAS_WORDS) /* NT 2: 36 KB, NT 3: 52 KB */
#define M3_CHUNK_PIECES_PER_WAVE(NT) (M3_CHUNK_WORDS(NT) / 256 / 4)  /* 9 / 13 */
#define M3_TOTAL_WORDS(NT) (M3_NCHUNKS * M3_CHUNK_WORDS(NT))
#define M3_LDS_BYTES(NT) (3 * M3_CHUNK_WORDS(NT) * 4)
#define M3_ROWS_PER_WG 128

/* one wave copies 1 KiB per instruction: lane i supplies bytes [16 i, 16 i + 16) */
template <int NT>
__device__ __forceinline__ void m3_stage(const uint32_t *w, uint32_t lds_slot_addr, int c, int wave, int lane) {
  const uint32_t *src = w + (size_t)c * M3_CHUNK_WORDS(NT) + lane * 4;
#pragma unroll
  for (int i = 0; i < M3_CHUNK_PIECES_PER_WAVE(NT); ++i) {
    const int p = wave + 4 * i;
    co_lds_dma_1k(src + p * 256, lds_slot_addr + (uint32_t)p * 1024u);
  }
}

/* (a, b) -> NT packed 16-bit pairs: the values rounded to bf16 (F16: fp16), then the successive remainders */
template <int NT, bool F16 = false>
__device__ __forceinline__ void m3_split(float a, float b, uint32_t (&t)[NT]) {
  if constexpr (F16 && NT == 2) {
    /* (as nn_rescnn.hip rcs_split: each second term is one mixed-precision fma, f16(a - float(t0.lo)); the difference is
     * exact in float32, the one rounding is the conversion's -- the same bits in three instructions instead of five) */
    const m3_f32x2 v2 = {a, b};
    const uint32_t t0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, m3_f16x2));
    uint32_t t1;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(t1)
        : "v"(t0), "v"(a), "v"(b));
    t[0] = t0;
    t[1] = t1;
    return;
  }
  m3_f32x2 v = {a, b};
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if constexpr (F16) {
      m3_f16x2 hb = __builtin_convertvector(v, m3_f16x2);
      t[i] = __builtin_bit_cast(uint32_t, hb);
      if (i + 1 < NT) {
        m3_f32x2 hf = __builtin_convertvector(hb, m3_f32x2);
        v = (m3_f32x2){v.x - hf.x, v.y - hf.y};
      }
    } else {
      m3_bf16x2 hb = __builtin_convertvector(v, m3_bf16x2);
      t[i] = __builtin_bit_cast(uint32_t, hb);
      if (i + 1 < NT) {
        m3_f32x2 hf = __builtin_convertvector(hb, m3_f32x2);
        v = (m3_f32x2){v.x - hf.x, v.y - hf.y};
      }
    }
  }
}

/* K steps S0 .. S0 + NS - 1 of a layer from one chunk: acc += W x.  XT = number of terms the B
 * operand has (1 for the input layer).  Products w_i x_j, i + j <= NT - 1, j < XT, largest
 * first.  Weight fragments of step s + 1 are requested from LDS before the MFMAs of step s issue. */
template <int NT, int S0, int NS, int XT, bool F16 = false>
__device__ __forceinline__ void m3_steps(m3_f32x16 (&acc)[4], const uint32_t (&b)[NT][8][4], const uint32_t *wl, int lane) {
  m3_u32x4 a[2][NT][4];
#pragma unroll
  for (int to = 0; to < 4; ++to)
#pragma unroll
    for (int t = 0; t < NT; ++t) a[0][t][to] = *reinterpret_cast<const m3_u32x4 *>(wl + (((0 * 4 + to) * NT + t) * 64 + lane) * 4);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < NS) {
#pragma unroll
      for (int to = 0; to < 4; ++to)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          a[nxt][t][to] = *reinterpret_cast<const m3_u32x4 *>(wl + ((((s + 1) * 4 + to) * NT + t) * 64 + lane) * 4);
    }
    m3_u32x4 B[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int m = 0; m < 4; ++m) B[t][m] = b[t][S0 + s][m];
    }
#pragma unroll
    for (int sum = 0; sum < NT; ++sum)
#pragma unroll
      for (int i = 0; i <= sum; ++i)
        if (sum - i < XT) {
#pragma unroll
          for (int to = 0; to < 4; ++to) {
            if constexpr (F16)
              acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(m3_f16x8, a[cur][i][to]),
                                                               __builtin_bit_cast(m3_f16x8, B[sum - i]), acc[to], 0, 0, 0);
            else
              acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(m3_bf16x8, a[cur][i][to]),
                                                                __builtin_bit_cast(m3_bf16x8, B[sum - i]), acc[to], 0, 0, 0);
          }
        }
  }
}

/* accumulators := bias (feature 32T + 8g + 4h + i in register 4g + i of tile T) */
__device__ __forceinline__ void m3_init_bias(m3_f32x16 (&acc)[4], const float *bias, int h) {
#pragma unroll
  for (int T = 0; T < 4; ++T)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *reinterpret_cast<const float4 *>(bias + 32 * T + 8 * g + 4 * h);
      acc[T][4 * g + 0] = b4.x;
      acc[T][4 * g + 1] = b4.y;
      acc[T][4 * g + 2] = b4.z;
      acc[T][4 * g + 3] = b4.w;
    }
}

/* top of chunk c: everything this wave requested except the youngest chunk has landed; after the
 * barrier that holds for every wave, and everyone has left the slot of chunk c - 1, which
 * receives chunk c + 2 */
#define M3_CHUNK_HEAD(NT, c)                                                                         \
  {                                                                                                  \
    if ((c) + 1 < M3_NCHUNKS) {                                                                      \
      if (NT == 2) CO_WAIT_VMCNT(9);                                                                 \
      else CO_WAIT_VMCNT(13);                                                                        \
    } else {                                                                                         \
      CO_WAIT_VMCNT(0);                                                                              \
    }                                                                                                \
    co_wg_barrier();                                                                                 \
    if ((c) + 2 < M3_NCHUNKS)                                                                        \
      m3_stage<NT>(wfrag, lds_base + (uint32_t)(((c) + 2) % 3) * (M3_CHUNK_WORDS(NT) * 4u), (c) + 2, wave, lane); \
  }

template <int NT, bool F16 = false>
__global__ __launch_bounds__(256, 1) void co_k_mlp_forward_split_t(const float *__restrict__ in, const int32_t *__restrict__ d_rows,
                                                                   const uint32_t *__restrict__ wfrag, float *__restrict__ eval,
                                                                   float *__restrict__ probs, CoNetIO io, uint32_t *range_flag) {
  static_assert(M3_CHUNK_PIECES_PER_WAVE(2) == 9 && M3_CHUNK_PIECES_PER_WAVE(3) == 13, "vmcnt immediates of M3_CHUNK_HEAD");
  extern __shared__ __attribute__((aligned(16))) uint32_t m3_lds[]; /* ring of three chunk slots */
  const int rows = *d_rows;
  const int row0 = blockIdx.x * M3_ROWS_PER_WG;
  if (row0 >= rows) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, h = lane >> 5, n = lane & 31;
  const uint32_t lds_base = co_lds_addr(m3_lds);

  /* inputs as the B operand of layer 0: step s = 2T + a, slot j <-> input 32T + 16a + 8(j/4) + 4h + j%4;
   * every input is 0, 1 or k/4: exact in bf16, so only the first term exists */
  const int row = row0 + wave * 32 + n;
  uint32_t b[NT][8][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m) b[t][s][m] = 0u;
  if (row < rows) {
    const float *x = in + (size_t)(io.in_idx ? io.in_idx[row] : row) * CO_STATE_STRIDE;
#pragma unroll
    for (int s = 0; s < M3_STEPS_L0; ++s) {
      const float4 v0 = *reinterpret_cast<const float4 *>(x + 16 * s + 4 * h);
      const float4 v1 = *reinterpret_cast<const float4 *>(x + 16 * s + 8 + 4 * h);
      uint32_t t[NT];
      m3_split<NT, F16>(v0.x, v0.y, t);
      b[0][s][0] = t[0];
      m3_split<NT, F16>(v0.z, v0.w, t);
      b[0][s][1] = t[0];
      m3_split<NT, F16>(v1.x, v1.y, t);
      b[0][s][2] = t[0];
      m3_split<NT, F16>(v1.z, v1.w, t);
      b[0][s][3] = t[0];
    }
  }
  /* the first two chunks are requested behind the input loads: vmcnt retires in issue order, so
   * loads issued behind a transfer would wait for it */
  m3_stage<NT>(wfrag, lds_base, 0, wave, lane);
  m3_stage<NT>(wfrag, lds_base + M3_CHUNK_WORDS(NT) * 4u, 1, wave, lane);
  m3_f32x16 acc[4];
  uint32_t amax = 0u; /* F16: the running maximum of the packed first terms (nn.h range_exceeded; nn_rescnn.hip rcs_pk_max_f16: an
                       * activation beyond fp16's range has the first term +inf; ReLU outputs are never negative) */
  for (int l = 0; l < M3_NLAYERS; ++l) {
    const int c0 = 2 * l;
    {
      M3_CHUNK_HEAD(NT, c0)
      const uint32_t *wl = m3_lds + (c0 % 3) * M3_CHUNK_WORDS(NT);
      m3_init_bias(acc, reinterpret_cast<const float *>(wl + 4 * M3_STEP_WORDS(NT)), h);
      if (l == 0) m3_steps<NT, 0, 4, 1, F16>(acc, b, wl, lane);
      else m3_steps<NT, 0, 4, NT, F16>(acc, b, wl, lane);
    }
    {
      M3_CHUNK_HEAD(NT, c0 + 1)
      const uint32_t *wl = m3_lds + ((c0 + 1) % 3) * M3_CHUNK_WORDS(NT);
      if (l == 0) m3_steps<NT, 4, M3_STEPS_L0 - 4, 1, F16>(acc, b, wl, lane);
      else m3_steps<NT, 4, M3_STEPS - 4, NT, F16>(acc, b, wl, lane);
    }
    if (l + 1 < M3_NLAYERS) {
      /* ReLU, then the term operands of the next layer: step 2T + a, word m = registers
       * (8a + 2m, 8a + 2m + 1) of tile T (BatchNorm lives in the next layer's weights) */
#pragma unroll
      for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            float v0 = acc[T][8 * a + 2 * m], v1 = acc[T][8 * a + 2 * m + 1];
            v0 = v0 > 0.0f ? v0 : 0.0f;
            v1 = v1 > 0.0f ? v1 : 0.0f;
            uint32_t t[NT];
            m3_split<NT, F16>(v0, v1, t);
            if constexpr (F16) asm("v_pk_max_f16 %0, %1, %2" : "=v"(amax) : "v"(amax), "v"(t[0]));
#pragma unroll
            for (int i = 0; i < NT; ++i) b[i][2 * T + a][m] = t[i];
          }
    }
  }
  if constexpr (F16) {
    if (!((amax & 0x7FFFu) < 0x7C00u && ((amax >> 16) & 0x7FFFu) < 0x7C00u)) atomicOr(range_flag, 1u); /* (never in range: no lane enters) */
  }
  /* heads: features 0..95 = policy logits (tiles 0..2), feature 96 = value (tile 3, g 0, h 0, i 0) */
  float mx = -INFINITY;
#pragma unroll
  for (int T = 0; T < 3; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = acc[T][r] > mx ? acc[T][r] : mx;
  float o = __shfl_xor(mx, 32, 64);
  mx = o > mx ? o : mx;
  float sum = 0.0f;
#pragma unroll
  for (int T = 0; T < 3; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[T][r] = __builtin_amdgcn_exp2f((acc[T][r] - mx) * 1.44269504088896340736f);
      sum += acc[T][r];
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (row < rows) {
    const size_t orow = (size_t)(io.out_idx ? io.out_idx[row] : row);
#pragma unroll
    for (int T = 0; T < 3; ++T)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 p = make_float4(acc[T][4 * g] * inv, acc[T][4 * g + 1] * inv, acc[T][4 * g + 2] * inv, acc[T][4 * g + 3] * inv);
        *reinterpret_cast<float4 *>(probs + orow * (size_t)io.probs_stride + 32 * T + 8 * g + 4 * h) = p;
      }
    if (h == 0) eval[orow * (size_t)io.eval_stride] = tanhf(acc[3][0]);
  }
}
#define co_k_mlp_forward_x3 co_k_mlp_forward_split_t<2>
#define co_k_mlp_forward_x6 co_k_mlp_forward_split_t<3>
/* "f16x3": two fp16 terms per operand (22 significand bits), three MFMA products -- see nn_rescnn.hip */
#define co_k_mlp_forward_h3 co_k_mlp_forward_split_t<2, true>

/* ------------------------------------------------------------------ host */
static inline uint16_t m3_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float m3_bf16_to_f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline uint16_t m3_f16_rne(float f) {
  _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static inline float m3_f16_to_f(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}

struct MlpSplitNet : CoNet {
  uint32_t *d_w = nullptr;
  uint32_t *d_range = nullptr; /* f16: the kernels' out-of-range flag */
  size_t cap;
  int nt;
  bool f16;
  MlpSplitNet(const float *w, size_t max_rows, rt_stream_t s, int nterms, bool fp16 = false) : cap(max_rows), nt(nterms), f16(fp16) {
    /* float64 copies of the 13 dense layers with BatchNorm l folded into layer l + 1 */
    std::vector<std::vector<double>> K(M3_NLAYERS), B(M3_NLAYERS);
    std::vector<int> kin(M3_NLAYERS), kout(M3_NLAYERS);
    const float *p = w;
    int in_dim = 70;
    std::vector<double> a_prev, c_prev;
    auto fold = [&](int l, const float *kern, const float *bias, int nin, int nout, int out_base) {
      /* K[l][k * 128 + out_base + o], B[l][out_base + o] */
      for (int o = 0; o < nout; ++o) {
        double b = bias[o];
        for (int k = 0; k < nin; ++k) {
          double wv = kern[(size_t)k * nout + o];
          if (!a_prev.empty()) {
            b += c_prev[k] * wv;
            wv *= a_prev[k];
          }
          K[l][(size_t)k * 128 + out_base + o] = wv;
        }
        B[l][out_base + o] = b;
      }
    };
    for (int l = 0; l < 12; ++l) {
      const float *kern = p, *b = kern + (size_t)in_dim * 100, *ga = b + 100, *be = ga + 100, *mu = be + 100, *va = mu + 100;
      K[l].assign((size_t)128 * 128, 0.0);
      B[l].assign(128, 0.0);
      kin[l] = in_dim;
      fold(l, kern, b, in_dim, 100, 0);
      a_prev.assign(100, 0.0);
      c_prev.assign(100, 0.0);
      for (int o = 0; o < 100; ++o) {
        /* the float32 constants K5 applies (BatchNormalization inference, eps 1e-3) */
        float a = (float)((double)ga[o] / sqrt((double)va[o] + CO_BN_EPS));
        a_prev[o] = a;
        c_prev[o] = (float)((double)be[o] - (double)mu[o] * (double)a);
      }
      p = va + 100;
      in_dim = 100;
    }
    const float *Kv = p, *bv = Kv + 100, *Kp = bv + 1, *bp = Kp + 9600;
    K[12].assign((size_t)128 * 128, 0.0);
    B[12].assign(128, 0.0);
    kin[12] = 100;
    fold(12, Kp, bp, 100, 96, 0);
    fold(12, Kv, bv, 100, 1, 96);
    const size_t step_words = (size_t)4 * nt * 256, chunk_words = 4 * step_words + M3_BIAS_WORDS;
    std::vector<uint32_t> buf((size_t)M3_NCHUNKS * chunk_words, 0u);
    for (int l = 0; l < M3_NLAYERS; ++l) {
      const int ns = l == 0 ? M3_STEPS_L0 : M3_STEPS;
      for (int st = 0; st < ns; ++st) {
        /* chunk 2l holds steps 0..3 (and the bias), chunk 2l + 1 steps 4.. */
        const size_t off = ((size_t)2 * l + (st >> 2)) * chunk_words + (size_t)(st & 3) * step_words;
        for (int to = 0; to < 4; ++to)
          for (int h = 0; h < 2; ++h)
            for (int i = 0; i < 32; ++i)
              for (int j = 0; j < 8; ++j) {
                /* step st = 2T + a; k-slot (h, j) <-> input feature 32T + 8(2a + j/4) + 4h + j%4 */
                int T = st >> 1, a = st & 1;
                int k = 32 * T + 8 * (2 * a + (j >> 2)) + 4 * h + (j & 3);
                int o = 32 * to + i;
                float v = k < kin[l] ? (float)K[l][(size_t)k * 128 + o] : 0.0f;
                if (f16 && !(fabsf(v) <= CO_F16_MAX))
                  throw std::invalid_argument("mlp12x100h3: a weight of layer " + std::to_string(l) + " is " + std::to_string(v) +
                                              " after the BatchNorm fold, beyond the fp16 range of the f16x3 kernels: use mlp12x100x6");
                size_t lane = 32 * h + i;
                for (int t = 0; t < nt; ++t) {
                  uint16_t term = f16 ? m3_f16_rne(v) : m3_bf16_rne(v);
                  v = v - (f16 ? m3_f16_to_f(term) : m3_bf16_to_f(term));
                  buf[off + (((size_t)to * nt + t) * 64 + lane) * 4 + j / 2] |= (uint32_t)term << (16 * (j & 1));
                }
              }
      }
      for (int o = 0; o < 128; ++o) {
        float b = (float)B[l][o];
        memcpy(&buf[(size_t)2 * l * chunk_words + 4 * step_words + o], &b, 4);
      }
    }
    rt_malloc((void **)&d_w, buf.size() * 4, s);
    rt_h2d(d_w, buf.data(), buf.size() * 4, s);
    if (f16) rt_malloc((void **)&d_range, 4, s);
    rt_sync(s);
    if (f16)
      RT_CHECK(hipFuncSetAttribute((const void *)co_k_mlp_forward_h3, hipFuncAttributeMaxDynamicSharedMemorySize, M3_LDS_BYTES(2)));
    else if (nt == 2)
      RT_CHECK(hipFuncSetAttribute((const void *)co_k_mlp_forward_x3, hipFuncAttributeMaxDynamicSharedMemorySize, M3_LDS_BYTES(2)));
    else
      RT_CHECK(hipFuncSetAttribute((const void *)co_k_mlp_forward_x6, hipFuncAttributeMaxDynamicSharedMemorySize, M3_LDS_BYTES(3)));
  }
  ~MlpSplitNet() override {
    rt_free(d_w);
    rt_free(d_range);
  }
  bool range_exceeded(rt_stream_t s) override {
    if (!d_range) return false;
    uint32_t flag = 0;
    rt_d2h(&flag, d_range, 4, s);
    rt_sync(s);
    return flag != 0;
  }
  size_t max_rows() const override { return cap; }
  int kind() const override { return f16 ? CO_NET_MLP12X100_H3 : nt == 2 ? CO_NET_MLP12X100_X3 : CO_NET_MLP12X100_X6; }
  double flop_per_row() const override { return 2.0 * (70 * 100 + 11 * 100 * 100 + 100 + 100 * 96); }
  void forward(const float *d_in, int32_t rows_cap, const int32_t *d_rows, float *d_eval, float *d_probs,
               rt_stream_t s, const CoNetIO &io = CoNetIO()) override {
    int grid = (rows_cap + M3_ROWS_PER_WG - 1) / M3_ROWS_PER_WG;
    if (grid < 1) return;
    if (f16)
      hipLaunchKernelGGL(co_k_mlp_forward_h3, dim3(grid), dim3(256), M3_LDS_BYTES(2), s, d_in, d_rows, (const uint32_t *)d_w, d_eval,
                         d_probs, io, d_range);
    else if (nt == 2)
      hipLaunchKernelGGL(co_k_mlp_forward_x3, dim3(grid), dim3(256), M3_LDS_BYTES(2), s, d_in, d_rows, (const uint32_t *)d_w, d_eval,
                         d_probs, io, d_range);
    else
      hipLaunchKernelGGL(co_k_mlp_forward_x6, dim3(grid), dim3(256), M3_LDS_BYTES(3), s, d_in, d_rows, (const uint32_t *)d_w, d_eval,
                         d_probs, io, d_range);
    RT_CHECK(hipGetLastError());
  }
};

CoNet *co_mlp_split_create(const float *weights, size_t n_floats, size_t max_rows, rt_stream_t s, int nterms, bool f16) {
  if (n_floats != (size_t)CO_MLP_NUM_WEIGHTS || (nterms != 2 && nterms != 3) || (f16 && nterms != 2)) return nullptr;
  return new MlpSplitNet(weights, max_rows, s, nterms, f16);
}
