// nn_mlp_x3.hip -- K5x3: the reference's policy/value network (12 x [Dense(100) -> ReLU ->
// BatchNorm] -> {Dense(1) tanh, Dense(96) softmax}, wrapper.py:256-271) as ONE fused gfx950
// kernel at bf16x3 split precision on v_mfma_f32_32x32x16_bf16.
//
// Same network, same flat weights and the same function (<= 1e-4) as K5 (nn_mlp.hip, fp32 MFMA);
// what changes is the arithmetic of the thirteen matrix products: x = hi + lo (two bf16),
// x w ~ hi hi + hi lo + lo hi with fp32 accumulation -- three bf16 MFMAs at 16x the fp32
// matrix rate.  The dense layers of this network are too small for fp32 MFMA to pay: K5 spends
// as long on them as the whole tree search takes.
//
// Mapping (the one of the residual CNN kernel, nn_rescnn.hip).  Transposed evaluation:
// out^T[feature][row] = W^T[feature][k] act^T[k][row]; a wave owns 32 batch rows = one MFMA
// column tile and all 128 (padded) output features = four 32-row tiles, 64 accumulator
// registers.  The accumulator layout (lane = (h, row), register 4g + i of tile T = feature
// 32T + 8g + 4h + i) is the B-operand layout of the next layer when its K steps are taken in the
// order  step s = 2T + a, k-slot (h, j) <-> feature 32T + 8(2a + j/4) + 4h + j%4,  so activations
// stay in registers through all 13 layers; the weights are pre-permuted (and pre-split into
// hi/lo bf16) on the host into that fragment order and stream through a double-buffered LDS
// window by LDS-DMA, one layer (57 KB) at a time.  BatchNorm of layer l is folded into the
// weights and bias of layer l + 1 on the host (in float64, as the TFLite converter does for the
// reference's own checkpoints), the bias is the initial value of the accumulators, so a layer's
// epilogue is ReLU + the hi/lo split.  tanh and the 96-way softmax are fused into the last
// layer.  Fixed k order, no batch-dependent tiling: a row's result does not depend on its batch.
//
// Geometry: 256 threads = 4 waves x 32 rows = 128 rows per workgroup, one
// workgroup per CU (2 x 58 KB of LDS); K = 112 (7 steps) for the hidden layers, 80 (5 steps) for
// the input layer, whose operands are exact in bf16 (0/1 and k/4) and need no lo product.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "engine_defs.h"
#include "nn.h"

typedef float m3_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 m3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 m3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float m3_f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t m3_u32x4 __attribute__((ext_vector_type(4)));

#define M3_NLAYERS 13
#define M3_STEPS 7      /* hidden layers and heads: K = 112 */
#define M3_STEPS_L0 5   /* input layer: K = 80 */
#define M3_STEP_WORDS (4 * 2 * 64 * 4) /* 4 output tiles x (hi, lo) x 64 lanes x 4 words */
#define M3_BIAS_WORDS 256              /* 128 biases, padded to one 1 KiB piece */
#define M3_CHUNK_WORDS (M3_STEPS * M3_STEP_WORDS + M3_BIAS_WORDS)       /* 14 592 words = 58 368 B */
#define M3_CHUNK0_WORDS (M3_STEPS_L0 * M3_STEP_WORDS + M3_BIAS_WORDS)   /* 10 496 words */
#define M3_TOTAL_WORDS (M3_CHUNK0_WORDS + 12 * M3_CHUNK_WORDS)
#define M3_LDS_BYTES (2 * M3_CHUNK_WORDS * 4)
#define M3_ROWS_PER_WG 128

__device__ __forceinline__ const uint32_t *m3_chunk_ptr(const uint32_t *w, int l) {
  return l == 0 ? w : w + M3_CHUNK0_WORDS + (size_t)(l - 1) * M3_CHUNK_WORDS;
}

/* one wave copies 1 KiB per instruction: lane i supplies bytes [16 i, 16 i + 16) */
__device__ __forceinline__ void m3_stage(const uint32_t *w, uint32_t *lds_buf, int l, int wave, int lane) {
  const uint32_t *src = m3_chunk_ptr(w, l);
  const int pieces = (l == 0 ? M3_CHUNK0_WORDS : M3_CHUNK_WORDS) / 256;
  for (int p = wave; p < pieces; p += 4)
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + p * 256 + lane * 4),
                                     (void __attribute__((address_space(3))) *)(lds_buf + p * 256), 16, 0, 0);
}

/* (a, b) -> packed bf16 pair of the values and packed bf16 pair of the remainders */
__device__ __forceinline__ void m3_split(float a, float b, uint32_t &hi, uint32_t &lo) {
  m3_f32x2 v = {a, b};
  m3_bf16x2 h = __builtin_convertvector(v, m3_bf16x2);
  m3_f32x2 hf = __builtin_convertvector(h, m3_f32x2);
  m3_f32x2 rem = {a - hf.x, b - hf.y};
  m3_bf16x2 l = __builtin_convertvector(rem, m3_bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

/* one layer: acc (initialised with the bias) += W x over NS K steps; LO = the B operand has a
 * lo part.  Weight fragments of step s + 1 are requested before the MFMAs of step s issue. */
template <int NS, bool LO>
__device__ __forceinline__ void m3_layer(m3_f32x16 (&acc)[4], const uint32_t (&bh)[8][4], const uint32_t (&bl)[8][4],
                                         const uint32_t *wl, int lane) {
  m3_u32x4 ah[2][4], al[2][4];
#pragma unroll
  for (int to = 0; to < 4; ++to) {
    ah[0][to] = *reinterpret_cast<const m3_u32x4 *>(wl + (((0 * 4 + to) * 2 + 0) * 64 + lane) * 4);
    al[0][to] = *reinterpret_cast<const m3_u32x4 *>(wl + (((0 * 4 + to) * 2 + 1) * 64 + lane) * 4);
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < NS) {
#pragma unroll
      for (int to = 0; to < 4; ++to) {
        ah[nxt][to] = *reinterpret_cast<const m3_u32x4 *>(wl + ((((s + 1) * 4 + to) * 2 + 0) * 64 + lane) * 4);
        al[nxt][to] = *reinterpret_cast<const m3_u32x4 *>(wl + ((((s + 1) * 4 + to) * 2 + 1) * 64 + lane) * 4);
      }
    }
    m3_u32x4 vh, vl;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      vh[m] = bh[s][m];
      vl[m] = bl[s][m];
    }
    const m3_bf16x8 Bh = __builtin_bit_cast(m3_bf16x8, vh), Bl = __builtin_bit_cast(m3_bf16x8, vl);
#pragma unroll
    for (int to = 0; to < 4; ++to)
      acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(m3_bf16x8, ah[cur][to]), Bh, acc[to], 0, 0, 0);
    if (LO) {
#pragma unroll
      for (int to = 0; to < 4; ++to)
        acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(m3_bf16x8, ah[cur][to]), Bl, acc[to], 0, 0, 0);
    }
#pragma unroll
    for (int to = 0; to < 4; ++to)
      acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(m3_bf16x8, al[cur][to]), Bh, acc[to], 0, 0, 0);
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

__global__ __launch_bounds__(256, 1) void co_k_mlp_forward_x3(const float *__restrict__ in, const int32_t *__restrict__ d_rows,
                                                              const uint32_t *__restrict__ wfrag, float *__restrict__ eval,
                                                              float *__restrict__ probs) {
  extern __shared__ __attribute__((aligned(16))) uint32_t m3_lds[]; /* two layer windows */
  const int rows = *d_rows;
  const int row0 = blockIdx.x * M3_ROWS_PER_WG;
  if (row0 >= rows) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, n = lane & 31;
  m3_stage(wfrag, m3_lds, 0, wave, lane);

  /* inputs as the B operand of layer 0: step s = 2T + a, slot j <-> input 32T + 16a + 8(j/4) + 4h + j%4;
   * every input is 0, 1 or k/4: exact in bf16, so there is no lo part */
  const int row = row0 + wave * 32 + n;
  uint32_t bh[8][4], bl[8][4];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int m = 0; m < 4; ++m) bh[s][m] = bl[s][m] = 0u;
  if (row < rows) {
    const float *x = in + (size_t)row * CO_STATE_STRIDE;
#pragma unroll
    for (int s = 0; s < M3_STEPS_L0; ++s) {
      const float4 v0 = *reinterpret_cast<const float4 *>(x + 16 * s + 4 * h);
      const float4 v1 = *reinterpret_cast<const float4 *>(x + 16 * s + 8 + 4 * h);
      uint32_t lo;
      m3_split(v0.x, v0.y, bh[s][0], lo);
      m3_split(v0.z, v0.w, bh[s][1], lo);
      m3_split(v1.x, v1.y, bh[s][2], lo);
      m3_split(v1.z, v1.w, bh[s][3], lo);
    }
  }
  m3_f32x16 acc[4];
  for (int l = 0; l < M3_NLAYERS; ++l) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads(); /* layer l has landed; everyone has left the other window */
    if (l + 1 < M3_NLAYERS) m3_stage(wfrag, m3_lds + ((l + 1) & 1) * M3_CHUNK_WORDS, l + 1, wave, lane);
    const uint32_t *wl = m3_lds + (l & 1) * M3_CHUNK_WORDS;
    if (l == 0) {
      m3_init_bias(acc, reinterpret_cast<const float *>(wl + M3_STEPS_L0 * M3_STEP_WORDS), h);
      m3_layer<M3_STEPS_L0, false>(acc, bh, bl, wl, lane);
    } else {
      m3_init_bias(acc, reinterpret_cast<const float *>(wl + M3_STEPS * M3_STEP_WORDS), h);
      m3_layer<M3_STEPS, true>(acc, bh, bl, wl, lane);
    }
    if (l + 1 < M3_NLAYERS) {
      /* ReLU, then the hi/lo operands of the next layer: step 2T + a, word m = registers
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
            m3_split(v0, v1, bh[2 * T + a][m], bl[2 * T + a][m]);
          }
    }
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
#pragma unroll
    for (int T = 0; T < 3; ++T)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 p = make_float4(acc[T][4 * g] * inv, acc[T][4 * g + 1] * inv, acc[T][4 * g + 2] * inv, acc[T][4 * g + 3] * inv);
        *reinterpret_cast<float4 *>(probs + (size_t)row * CO_NUM_MOVES + 32 * T + 8 * g + 4 * h) = p;
      }
    if (h == 0) eval[row] = tanhf(acc[3][0]);
  }
}

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

struct MlpX3Net : CoNet {
  uint32_t *d_w = nullptr;
  size_t cap;
  MlpX3Net(const float *w, size_t max_rows, rt_stream_t s) : cap(max_rows) {
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
    std::vector<uint32_t> buf(M3_TOTAL_WORDS, 0u);
    size_t off = 0;
    for (int l = 0; l < M3_NLAYERS; ++l) {
      const int ns = l == 0 ? M3_STEPS_L0 : M3_STEPS;
      for (int st = 0; st < ns; ++st)
        for (int to = 0; to < 4; ++to)
          for (int h = 0; h < 2; ++h)
            for (int i = 0; i < 32; ++i)
              for (int j = 0; j < 8; ++j) {
                /* step st = 2T + a; k-slot (h, j) <-> input feature 32T + 8(2a + j/4) + 4h + j%4 */
                int T = st >> 1, a = st & 1;
                int k = 32 * T + 8 * (2 * a + (j >> 2)) + 4 * h + (j & 3);
                int o = 32 * to + i;
                float v = k < kin[l] ? (float)K[l][(size_t)k * 128 + o] : 0.0f;
                uint16_t hi = m3_bf16_rne(v);
                uint16_t lo = m3_bf16_rne(v - m3_bf16_to_f(hi));
                size_t lane = 32 * h + i;
                size_t wh = off + ((((size_t)st * 4 + to) * 2 + 0) * 64 + lane) * 4 + j / 2;
                size_t wl = off + ((((size_t)st * 4 + to) * 2 + 1) * 64 + lane) * 4 + j / 2;
                buf[wh] |= (uint32_t)hi << (16 * (j & 1));
                buf[wl] |= (uint32_t)lo << (16 * (j & 1));
              }
      for (int o = 0; o < 128; ++o) {
        float b = (float)B[l][o];
        memcpy(&buf[off + (size_t)ns * M3_STEP_WORDS + o], &b, 4);
      }
      off += (size_t)ns * M3_STEP_WORDS + M3_BIAS_WORDS;
    }
    rt_malloc((void **)&d_w, buf.size() * 4);
    rt_h2d(d_w, buf.data(), buf.size() * 4, s);
    rt_sync(s);
    RT_CHECK(hipFuncSetAttribute((const void *)co_k_mlp_forward_x3, hipFuncAttributeMaxDynamicSharedMemorySize, M3_LDS_BYTES));
  }
  ~MlpX3Net() override { rt_free(d_w); }
  size_t max_rows() const override { return cap; }
  int kind() const override { return CO_NET_MLP12X100_X3; }
  double flop_per_row() const override { return 2.0 * (70 * 100 + 11 * 100 * 100 + 100 + 100 * 96); }
  void forward(const float *d_in, int32_t rows_cap, const int32_t *d_rows, float *d_eval, float *d_probs,
               rt_stream_t s) override {
    int grid = (rows_cap + M3_ROWS_PER_WG - 1) / M3_ROWS_PER_WG;
    if (grid < 1) return;
    hipLaunchKernelGGL(co_k_mlp_forward_x3, dim3(grid), dim3(256), M3_LDS_BYTES, s, d_in, d_rows, (const uint32_t *)d_w, d_eval,
                       d_probs);
    RT_CHECK(hipGetLastError());
  }
};

CoNet *co_mlp_x3_create(const float *weights, size_t n_floats, size_t max_rows, rt_stream_t s) {
  if (n_floats != (size_t)CO_MLP_NUM_WEIGHTS) return nullptr;
  return new MlpX3Net(weights, max_rows, s);
}
