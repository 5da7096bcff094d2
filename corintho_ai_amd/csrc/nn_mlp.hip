// nn_mlp.hip -- K5: the reference's policy/value network (12 x [Dense(100) ->
// ReLU -> BatchNorm] -> {Dense(1) tanh, Dense(96) softmax}, wrapper.py:256-271)
// as ONE fused gfx950 kernel on fp32 MFMA.
//
// Mapping.  The network is evaluated transposed: out^T[feature][row] =
// W^T[feature][k] * act^T[k][row], with v_mfma_f32_16x16x4_f32.  A 16x16 output
// tile keeps its batch row on the lane (col = lane & 15) and four features in
// the four accumulator registers (row = 4*(lane>>4) + reg) -- which is exactly
// the B-operand layout (B[k = lane>>4][col = lane&15]) of the next layer if its
// K steps are taken in the order k(step = 4*tile + reg, lane group q) =
// 16*tile + 4*q + reg.  So activations never leave registers between the 13
// layers; only the weights move, pre-permuted on the host into that fragment
// order and staged per layer through LDS (28 steps x 7 tiles x 64 lanes x 4 B =
// 50 176 B).  Bias + ReLU + BatchNorm (inference affine) are applied to the
// accumulators in place; tanh and the 96-way softmax are fused into the last
// layer's epilogue.  fp32 end to end (the "exact f32" MFMA, one rounding per
// product, fixed k order), so a row's result does not depend on the batch.
//
// Work: 2*(70*100 + 11*100*100 + 100*97) = 253.4 KFLOP per row (algorithmic);
// the padded tiles (112x112) issue 28% more MFMA work than that.
// Geometry: 256 threads = 4 waves per workgroup, 32 rows per wave (two 16-row
// tiles share every weight fragment), 128 rows per workgroup.  The weights of
// layer l+1 stream into the second LDS buffer with global_load_lds (LDS-DMA, no
// registers) while layer l computes: 2 x 50 176 B of LDS, one workgroup per CU,
// one barrier per layer.
#include <hip/hip_runtime.h>
#include <math.h>

#include <vector>

#include "engine_defs.h"
#include "nn.h"

#define MLP_TILES 7               /* 7 x 16 = 112 >= 100 features */
#define MLP_STEPS 28              /* 7 tiles x 4 k-steps */
#define MLP_STEPS_L0 20           /* 80 padded inputs */
#define MLP_FRAG (MLP_TILES * 64) /* floats per step */
#define MLP_LAYER_FLOATS (MLP_STEPS * MLP_FRAG)
#define MLP_NLAYERS 13            /* 12 hidden + heads */
#define MLP_PADW 112
#define MLP_ROWS_PER_WG 128

typedef float f32x4 __attribute__((ext_vector_type(4)));

/* one wave copies 1 KiB per instruction: lane i supplies bytes [16 i, 16 i + 16) */
__device__ __forceinline__ void mlp_stage_weights(const float *__restrict__ src, float *lds_dst, int nsteps, int wave,
                                                  int lane) {
  const int chunks = nsteps * MLP_FRAG / 256; /* 1 KiB chunks: 49 for 28 steps, 35 for 20 */
  for (int ch = wave; ch < chunks; ch += 4) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + ch * 256 + lane * 4),
                                     (void __attribute__((address_space(3))) *)(lds_dst + ch * 256), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256, 1) void co_k_mlp_forward(const float *__restrict__ in, const int32_t *__restrict__ d_rows,
                                                           const float *__restrict__ wfrag,
                                                           const float *__restrict__ bias,
                                                           const float *__restrict__ bn_a,
                                                           const float *__restrict__ bn_b, float *__restrict__ eval,
                                                           float *__restrict__ probs, CoNetIO io) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[]; /* 2 x MLP_LAYER_FLOATS */
  const int rows = *d_rows;
  const int row0 = blockIdx.x * MLP_ROWS_PER_WG;
  if (row0 >= rows) return;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, c = lane & 15;
  mlp_stage_weights(wfrag, lds_all, MLP_STEPS_L0, wave, lane);

  float x[2][MLP_TILES][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int row = row0 + wave * 32 + nb * 16 + c;
    const bool valid = row < rows;
#pragma unroll
    for (int tt = 0; tt < MLP_TILES; ++tt) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tt < 5 && valid)
        v = *reinterpret_cast<const float4 *>(in + (size_t)(io.in_idx ? io.in_idx[row] : row) * CO_STATE_STRIDE + 16 * tt + 4 * q);
      x[nb][tt][0] = v.x;
      x[nb][tt][1] = v.y;
      x[nb][tt][2] = v.z;
      x[nb][tt][3] = v.w;
    }
  }

  for (int l = 0; l < MLP_NLAYERS; ++l) {
    const int nsteps = l == 0 ? MLP_STEPS_L0 : MLP_STEPS;
    const float *lds_w = lds_all + (l & 1) * MLP_LAYER_FLOATS;
    /* this wave's DMA pieces of layer l have landed; the barrier makes every wave's
     * pieces visible and proves nobody still reads the other buffer (layer l-1) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (l + 1 < MLP_NLAYERS)
      mlp_stage_weights(wfrag + (size_t)(l + 1) * MLP_LAYER_FLOATS, lds_all + ((l + 1) & 1) * MLP_LAYER_FLOATS, MLP_STEPS,
                        wave, lane);
    f32x4 acc[2][MLP_TILES];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int t = 0; t < MLP_TILES; ++t) acc[nb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tt = 0; tt < MLP_TILES; ++tt) {
      if (tt * 4 < nsteps) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float *wp = lds_w + (tt * 4 + r) * MLP_FRAG + lane;
          float a[MLP_TILES];
#pragma unroll
          for (int t = 0; t < MLP_TILES; ++t) a[t] = wp[t * 64];
#pragma unroll
          for (int t = 0; t < MLP_TILES; ++t) {
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], x[0][tt][r], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], x[1][tt][r], acc[1][t], 0, 0, 0);
          }
        }
      }
    }
    if (l < MLP_NLAYERS - 1) {
      /* Dense bias -> ReLU -> BatchNorm affine, feature f = 16t + 4q + r */
#pragma unroll
      for (int t = 0; t < MLP_TILES; ++t) {
        const int f = l * MLP_PADW + 16 * t + 4 * q;
        const float4 b4 = *reinterpret_cast<const float4 *>(bias + f);
        const float4 a4 = *reinterpret_cast<const float4 *>(bn_a + f);
        const float4 c4 = *reinterpret_cast<const float4 *>(bn_b + f);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
        const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
        const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[nb][t][r] + bb[r];
            v = v > 0.0f ? v : 0.0f;
            x[nb][t][r] = aa[r] * v + cc[r];
          }
      }
    } else {
      /* heads: features 0..95 policy logits, feature 96 (tile 6, q 0, r 0) value */
      const float bv = bias[l * MLP_PADW + 96];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int row = row0 + wave * 32 + nb * 16 + c;
        float lg[6][4];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          const float4 b4 = *reinterpret_cast<const float4 *>(bias + l * MLP_PADW + 16 * t + 4 * q);
          const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            lg[t][r] = acc[nb][t][r] + bb[r];
            m = lg[t][r] > m ? lg[t][r] : m;
          }
        }
        float o = __shfl_xor(m, 16, 64);
        m = o > m ? o : m;
        o = __shfl_xor(m, 32, 64);
        m = o > m ? o : m;
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            lg[t][r] = expf(lg[t][r] - m);
            s += lg[t][r];
          }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (row < rows) {
          const size_t orow = (size_t)(io.out_idx ? io.out_idx[row] : row);
#pragma unroll
          for (int t = 0; t < 6; ++t) {
            float4 p = make_float4(lg[t][0] / s, lg[t][1] / s, lg[t][2] / s, lg[t][3] / s);
            *reinterpret_cast<float4 *>(probs + orow * (size_t)io.probs_stride + 16 * t + 4 * q) = p;
          }
          if (q == 0) eval[orow * (size_t)io.eval_stride] = tanhf(acc[nb][6][0] + bv);
        }
      }
    }
  }
}

/* ------------------------------------------------------------------ host */
struct MlpNet : CoNet {
  float *d_wfrag = nullptr, *d_bias = nullptr, *d_a = nullptr, *d_b = nullptr;
  size_t cap;
  MlpNet(const float *w, size_t max_rows, rt_stream_t s) : cap(max_rows) {
    std::vector<float> wf((size_t)MLP_NLAYERS * MLP_LAYER_FLOATS, 0.0f);
    std::vector<float> bias((size_t)MLP_NLAYERS * MLP_PADW, 0.0f), ba((size_t)12 * MLP_PADW, 0.0f),
        bb((size_t)12 * MLP_PADW, 0.0f);
    const float *p = w;
    int in_dim = 70;
    auto put = [&](int l, int k, int o, float v) {
      /* wfrag[l][step = 4*tt + r][tile][lane = 16*q + i] = W[k = 16tt + 4q + r][o = 16*tile + i] */
      int tt = k / 16, q = (k % 16) / 4, r = k % 4;
      int tile = o / 16, i = o % 16;
      wf[(size_t)l * MLP_LAYER_FLOATS + ((size_t)(4 * tt + r) * MLP_TILES + tile) * 64 + 16 * q + i] = v;
    };
    for (int l = 0; l < 12; ++l) {
      const float *K = p, *b = K + (size_t)in_dim * 100, *ga = b + 100, *be = ga + 100, *mu = be + 100, *va = mu + 100;
      for (int k = 0; k < in_dim; ++k)
        for (int o = 0; o < 100; ++o) put(l, k, o, K[(size_t)k * 100 + o]);
      for (int o = 0; o < 100; ++o) {
        bias[(size_t)l * MLP_PADW + o] = b[o];
        /* BatchNormalization inference: gamma (x - mean) / sqrt(var + eps) + beta */
        float a = (float)((double)ga[o] / sqrt((double)va[o] + CO_BN_EPS));
        ba[(size_t)l * MLP_PADW + o] = a;
        bb[(size_t)l * MLP_PADW + o] = (float)((double)be[o] - (double)mu[o] * (double)a);
      }
      p = va + 100;
      in_dim = 100;
    }
    const float *Kv = p, *bv = Kv + 100, *Kp = bv + 1, *bp = Kp + 9600;
    for (int k = 0; k < 100; ++k) {
      for (int o = 0; o < 96; ++o) put(12, k, o, Kp[(size_t)k * 96 + o]);
      put(12, k, 96, Kv[k]);
    }
    for (int o = 0; o < 96; ++o) bias[(size_t)12 * MLP_PADW + o] = bp[o];
    bias[(size_t)12 * MLP_PADW + 96] = bv[0];
    rt_malloc((void **)&d_wfrag, wf.size() * 4, s);
    rt_malloc((void **)&d_bias, bias.size() * 4, s);
    rt_malloc((void **)&d_a, ba.size() * 4, s);
    rt_malloc((void **)&d_b, bb.size() * 4, s);
    rt_h2d(d_wfrag, wf.data(), wf.size() * 4, s);
    rt_h2d(d_bias, bias.data(), bias.size() * 4, s);
    rt_h2d(d_a, ba.data(), ba.size() * 4, s);
    rt_h2d(d_b, bb.data(), bb.size() * 4, s);
    rt_sync(s);
    RT_CHECK(hipFuncSetAttribute((const void *)co_k_mlp_forward, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 2 * MLP_LAYER_FLOATS * (int)sizeof(float)));
  }
  ~MlpNet() override {
    rt_free(d_wfrag);
    rt_free(d_bias);
    rt_free(d_a);
    rt_free(d_b);
  }
  size_t max_rows() const override { return cap; }
  int kind() const override { return CO_NET_MLP12X100; }
  double flop_per_row() const override { return 2.0 * (70 * 100 + 11 * 100 * 100 + 100 + 100 * 96); }
  void forward(const float *d_in, int32_t rows_cap, const int32_t *d_rows, float *d_eval, float *d_probs,
               rt_stream_t s, const CoNetIO &io = CoNetIO()) override {
    int grid = (rows_cap + MLP_ROWS_PER_WG - 1) / MLP_ROWS_PER_WG;
    if (grid < 1) return;
    hipLaunchKernelGGL(co_k_mlp_forward, dim3(grid), dim3(256), 2 * MLP_LAYER_FLOATS * sizeof(float), s, d_in, d_rows, (const float *)d_wfrag,
                       (const float *)d_bias, (const float *)d_a, (const float *)d_b, d_eval, d_probs, io);
    RT_CHECK(hipGetLastError());
  }
};

CoNet *co_rescnn_create(const float *weights, size_t n_floats, size_t max_rows, rt_stream_t s);
CoNet *co_rescnn_split_create(const float *weights, size_t n_floats, size_t max_rows, rt_stream_t s, int nterms, bool f16 = false);
CoNet *co_mlp_split_create(const float *weights, size_t n_floats, size_t max_rows, rt_stream_t s, int nterms, bool f16 = false);

CoNet *co_net_create(int kind, const float *weights, size_t n_floats, size_t max_rows, rt_stream_t s) {
  if (kind == CO_NET_MLP12X100 && n_floats == (size_t)CO_MLP_NUM_WEIGHTS) return new MlpNet(weights, max_rows, s);
  if (kind == CO_NET_RESCNN4) return co_rescnn_create(weights, n_floats, max_rows, s);
  if (kind == CO_NET_RESCNN4_X3) return co_rescnn_split_create(weights, n_floats, max_rows, s, 2);
  if (kind == CO_NET_RESCNN4_X6) return co_rescnn_split_create(weights, n_floats, max_rows, s, 3);
  if (kind == CO_NET_MLP12X100_X3) return co_mlp_split_create(weights, n_floats, max_rows, s, 2);
  if (kind == CO_NET_MLP12X100_X6) return co_mlp_split_create(weights, n_floats, max_rows, s, 3);
  if (kind == CO_NET_RESCNN4_H3) return co_rescnn_split_create(weights, n_floats, max_rows, s, 2, true);
  if (kind == CO_NET_MLP12X100_H3) return co_mlp_split_create(weights, n_floats, max_rows, s, 2, true);
  return nullptr;
}
