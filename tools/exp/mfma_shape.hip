// Experiment: which fp16 MFMA shape should the residual-CNN kernel's convolution loop use?
// The chip lowers its clock under an MFMA-dense loop on random data (MI355X_MICROARCH.md, DVFS give-back item 7), and the clock
// it holds can differ by shape.  Two stripped copies of the inner loop of nn_rescnn.hip rcs_conv_group (weights re-read from LDS
// by ds_read_b128, activations shifted by DPP, three products per operand pair, two waves per SIMD), same FLOP per wave:
//   A  v_mfma_f32_32x32x16_f16, 2 position pairs per wave  (the kernel as it is)
//   B  v_mfma_f32_16x16x32_f16, 4 positions per wave
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_shape.hip -o build_ab/mfma_shape && build_ab/mfma_shape
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GROUP_WORDS (3 * 4 * 2 * 2 * 64 * 4) /* taps x K=16 steps x tiles x terms x lanes x words = 48 KB, as in the kernel */

__device__ __forceinline__ uint32_t shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x111, 0xF, 0xF, true); }

struct Stamp {
  unsigned long long core, real;
};

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void conv_loop(const uint32_t *w, const uint32_t *x, float *out, int groups, Stamp *stamps) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2 * GROUP_WORDS; i += 512) lds[i] = w[i];
  __syncthreads();
  uint32_t pk[2][4][2][4]; /* term, position (A: pair np = pos / 2 .. the same 64 registers), K block, word */
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) pk[t][p][s][m] = x[((((blockIdx.x * 8 + (tid >> 6)) * 2 + t) * 4 + p) * 2 + s) * 256 + m * 64 + lane];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (SHAPE == 32) {
    f32x16 acc[2][2];
#pragma unroll
    for (int np = 0; np < 2; ++np)
#pragma unroll
      for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[np][to][i] = 0.0f;
    for (int g = 0; g < groups; ++g) {
      const uint32_t *wg = lds + (g & 1) * GROUP_WORDS;
#pragma unroll
      for (int idx = 0; idx < 12; ++idx) { /* 3 taps x 4 K steps of 16 */
        u32x4 a[2][2];
#pragma unroll
        for (int to = 0; to < 2; ++to)
#pragma unroll
          for (int t = 0; t < 2; ++t) a[t][to] = *reinterpret_cast<const u32x4 *>(wg + (((idx * 2 + to) * 2 + t) * 64 + lane) * 4);
#pragma unroll
        for (int np = 0; np < 2; ++np) {
          u32x4 B[2];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) B[t][m] = shr1(pk[t][2 * np + ((idx >> 1) & 1)][idx & 1][m]);
#pragma unroll
          for (int sum = 0; sum < 2; ++sum)
#pragma unroll
            for (int i = 0; i <= sum; ++i)
#pragma unroll
              for (int to = 0; to < 2; ++to)
                acc[np][to] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][to]), __builtin_bit_cast(f16x8, B[sum - i]),
                                                                     acc[np][to], 0, 0, 0);
        }
      }
    }
    float s = 0.0f;
#pragma unroll
    for (int np = 0; np < 2; ++np)
#pragma unroll
      for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[np][to][i];
    out[blockIdx.x * 512 + tid] = s;
  } else {
    f32x4 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[p][mt][i] = 0.0f;
    for (int g = 0; g < groups; ++g) {
      const uint32_t *wg = lds + (g & 1) * GROUP_WORDS;
#pragma unroll
      for (int idx = 0; idx < 6; ++idx) { /* 3 taps x 2 K steps of 32 */
        u32x4 a[2][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int t = 0; t < 2; ++t) a[t][mt] = *reinterpret_cast<const u32x4 *>(wg + (((idx * 4 + mt) * 2 + t) * 64 + lane) * 4);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          u32x4 B[2];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) B[t][m] = shr1(pk[t][p][idx & 1][m]);
#pragma unroll
          for (int sum = 0; sum < 2; ++sum)
#pragma unroll
            for (int i = 0; i <= sum; ++i)
#pragma unroll
              for (int mt = 0; mt < 4; ++mt)
                acc[p][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i][mt]), __builtin_bit_cast(f16x8, B[sum - i]),
                                                                    acc[p][mt], 0, 0, 0);
        }
      }
    }
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[p][mt][i];
    out[blockIdx.x * 512 + tid] = s;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    stamps[blockIdx.x].core = c1 - c0;
    stamps[blockIdx.x].real = r1 - r0;
  }
}

static uint16_t f16_bits(float f) {
  _Float16 h = (_Float16)f;
  uint16_t u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}

int main(int argc, char **argv) {
  const int wgs = 256 * 3, groups = 24 * 8; /* eight networks' worth of 64-channel groups per workgroup */
  std::vector<uint32_t> hw(2 * GROUP_WORDS), hx((size_t)wgs * 8 * 2 * 4 * 2 * 256);
  srand(1);
  auto rnd = [](float scale) { return scale * ((float)rand() / (float)RAND_MAX - 0.5f); };
  for (auto &v : hw) v = (uint32_t)f16_bits(rnd(0.2f)) | ((uint32_t)f16_bits(rnd(0.2f)) << 16);
  for (auto &v : hx) v = (uint32_t)f16_bits(rnd(2.0f)) | ((uint32_t)f16_bits(rnd(2.0f)) << 16);
  uint32_t *dw, *dx;
  float *dout;
  Stamp *ds;
  hipMalloc(&dw, hw.size() * 4);
  hipMalloc(&dx, hx.size() * 4);
  hipMalloc(&dout, (size_t)wgs * 512 * 4);
  hipMalloc(&ds, (size_t)wgs * sizeof(Stamp));
  hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  const size_t lds_bytes = 2 * GROUP_WORDS * 4 + 55 * 1024; /* 151 KB as the kernel: one workgroup per CU */
  hipFuncSetAttribute((const void *)conv_loop<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipFuncSetAttribute((const void *)conv_loop<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double flop = (double)wgs * 8 * groups * 144 * 32768.0 * 2 / 2; /* per launch: 144 MFMAs of 32x32x16 per wave and group */
  for (int round = 0; round < 2; ++round)
    for (int shape : {32, 16}) {
      const int reps = 400;
      auto launch = [&]() {
        if (shape == 32)
          hipLaunchKernelGGL(conv_loop<32>, dim3(wgs), dim3(512), lds_bytes, 0, dw, dx, dout, groups, ds);
        else
          hipLaunchKernelGGL(conv_loop<16>, dim3(wgs), dim3(512), lds_bytes, 0, dw, dx, dout, groups, ds);
      };
      for (int i = 0; i < reps; ++i) launch(); /* more than two seconds of back-to-back launches first */
      hipEventRecord(e0, 0);
      for (int i = 0; i < reps; ++i) launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<Stamp> hs(wgs);
      hipMemcpy(hs.data(), ds, wgs * sizeof(Stamp), hipMemcpyDeviceToHost);
      std::vector<double> clk, cyc;
      for (auto &s : hs) {
        clk.push_back((double)s.core / (double)s.real * 0.1);
        cyc.push_back((double)s.core);
      }
      std::sort(clk.begin(), clk.end());
      std::sort(cyc.begin(), cyc.end());
      printf("shape %s: %.3f ms per launch, %.1f TFLOP/s executed; in-kernel clock %.2f GHz (median), %.0f core cycles per workgroup pass "
             "(%.1f per MFMA-32 equivalent and SIMD)\n",
             shape == 32 ? "32x32x16" : "16x16x32", ms / reps, flop * reps / (ms * 1e-3) / 1e12, clk[wgs / 2], cyc[wgs / 2],
             cyc[wgs / 2] / (2.0 * groups * 144));
    }
  return 0;
}
