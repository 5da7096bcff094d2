// Diagnostic microbenchmark: the inner loop of the split-precision convolution (LDS weight-fragment reads +
// DPP operand shifts + bf16 MFMAs, two waves per SIMD, random data) with the 32x32x16 shape against the
// 16x16x32 shape, same FLOPs per k-step.  MI355X_MICROARCH.md (DVFS give-back, item 7) reports the 16x16x32
// shape holding a higher clock under load.  usage: mfma_shape [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x101, 0xF, 0xF, true); }

// one "k-step" of the 32x32x16 mapping: 6 weight fragments (2 tiles x 3 terms), 3 B operands, 12 MFMAs
__global__ __launch_bounds__(512, 2) void k32(const uint32_t *w, float *out, int iters) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 6 * 256 * 4; i += 512) lds[i] = w[i];
  __syncthreads();
  f32x16 acc[2];
  for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  u32x4 b[3];
  for (int t = 0; t < 3; ++t) for (int m = 0; m < 4; ++m) b[t][m] = w[(t * 4 + m) * 64 + lane];
  for (int it = 0; it < iters; ++it) {
    u32x4 a[3][2];
    for (int t = 0; t < 3; ++t) for (int to = 0; to < 2; ++to) a[t][to] = *(const u32x4 *)(lds + (((it & 3) * 6 + t * 2 + to) % 24 * 64 + lane) * 4);
    bf16x8 B[3];
    for (int t = 0; t < 3; ++t) { u32x4 s; for (int m = 0; m < 4; ++m) s[m] = shl1(b[t][m]); B[t] = __builtin_bit_cast(bf16x8, s); }
    for (int sum = 0; sum < 3; ++sum) for (int i = 0; i <= sum; ++i) for (int to = 0; to < 2; ++to)
      acc[to] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][to]), B[sum - i], acc[to], 0, 0, 0);
  }
  float s = 0; for (int t = 0; t < 2; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// the same FLOPs with 16x16x32: 4 tiles x 3 terms = 12 fragments, 2 positions x 3 B operands, 48 MFMAs -- for TWO
// k-steps of the loop above (K = 32), so per launch iteration count is halved by the caller
__global__ __launch_bounds__(512, 2) void k16(const uint32_t *w, float *out, int iters) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 6 * 256 * 4; i += 512) lds[i] = w[i];
  __syncthreads();
  f32x4 acc[2][4];
  for (int p = 0; p < 2; ++p) for (int t = 0; t < 4; ++t) for (int i = 0; i < 4; ++i) acc[p][t][i] = 0.f;
  u32x4 b[2][3];
  for (int p = 0; p < 2; ++p) for (int t = 0; t < 3; ++t) for (int m = 0; m < 4; ++m) b[p][t][m] = w[((p * 3 + t) * 4 + m) * 64 + lane];
  for (int it = 0; it < iters; ++it) {
    u32x4 a[3][4];
    for (int t = 0; t < 3; ++t) for (int to = 0; to < 4; ++to) a[t][to] = *(const u32x4 *)(lds + (((it & 1) * 12 + t * 4 + to) % 24 * 64 + lane) * 4);
    for (int p = 0; p < 2; ++p) {
      bf16x8 B[3];
      for (int t = 0; t < 3; ++t) { u32x4 s; for (int m = 0; m < 4; ++m) s[m] = shl1(b[p][t][m]); B[t] = __builtin_bit_cast(bf16x8, s); }
      for (int sum = 0; sum < 3; ++sum) for (int i = 0; i <= sum; ++i) for (int to = 0; to < 4; ++to)
        acc[p][to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i][to]), B[sum - i], acc[p][to], 0, 0, 0);
    }
  }
  float s = 0; for (int p = 0; p < 2; ++p) for (int t = 0; t < 4; ++t) for (int i = 0; i < 4; ++i) s += acc[p][t][i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  std::vector<uint32_t> h(6 * 256 * 4);
  srand(1);
  for (auto &x : h) { uint32_t a = 0x3f00 + (rand() & 0xff), b = 0xbf00 + (rand() & 0xff); x = (a << 16) | b; }  // random bf16 pairs ~ +-0.5..1
  uint32_t *dw; float *dout;
  hipMalloc(&dw, h.size() * 4); hipMalloc(&dout, 1024 * 512 * 4);
  hipMemcpy(dw, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256;  // one 512-thread workgroup per CU, two waves per SIMD
  for (int rep = 0; rep < 3; ++rep) {
    for (int which = 0; which < 2; ++which) {
      hipEventRecord(e0);
      for (int r = 0; r < 10; ++r) {
        if (which == 0) hipLaunchKernelGGL(k32, dim3(grid), dim3(512), 6 * 256 * 16, 0, dw, dout, iters);
        else hipLaunchKernelGGL(k16, dim3(grid), dim3(512), 6 * 256 * 16, 0, dw, dout, iters / 2);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // FLOPs: k32: iters x 12 MFMA x 32*32*16*2 per wave; k16: iters/2 x 48 x 16*16*32*2 -- equal
      double flop = 10.0 * grid * 8.0 * iters * 12.0 * 32 * 32 * 16 * 2;
      printf("%s: %.2f ms, %.1f TFLOP/s issued\n", which == 0 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9);
    }
  }
  return 0;
}
