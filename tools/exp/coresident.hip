// Diagnostic microbenchmark (round 6): can search-like wavefronts and MFMA wavefronts share a CU -- a
// SIMD -- and issue side by side?  Two kernels shaped after the two families of the self-play engine:
//   k_search   8 wavefronts per workgroup, <= 128 registers, 63 KB of LDS: every wavefront a CHAIN of dependent vector
//              instructions (double-precision fma, DPP maxima, v_readlane -> scalar use, LDS round trips) with a dependent
//              global load (pointer chase in a 1 GB buffer) every ~120 instructions -- the search kernel's profile: one
//              instruction per ~11 cycles, 40 % of the time at s_waitcnt;
//   k_mfma     4 wavefronts per workgroup (one per SIMD), <= 240 registers, 96 KB of LDS: v_mfma_f32_32x32x16_f16 back to
//              back on four accumulators with the B fragments re-read from LDS -- the network kernel's inner loop.
// One workgroup of each fits a CU together (159 KB of LDS; 2 x 128 + 240 registers per SIMD; 12 wavefronts), two of the same
// kind do not beside one of the other.  Each kernel runs alone and then both at once on two streams, (a) on the whole chip and
// (b) with both streams masked to the SAME compute units, with as many workgroups as there are units.  If the two kinds of
// wavefronts issue side by side, both together take as long as the longer one; if a SIMD's issue port is what they share,
// as long as the sum.
// usage: coresident [search_iters] [mfma_iters]
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_search(const uint32_t *chase, uint32_t mask, int iters, double *out) {
  __shared__ uint32_t lds[63 * 256]; /* 63 KB */
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *mine = lds + wave * 1024;
  uint32_t p = (blockIdx.x * 8u + wave) * 2654435761u & mask;
  double a = 1.0 + lane, b = 0.5;
  float m = (float)lane;
  for (int it = 0; it < iters; ++it) {
    const uint32_t nxt = chase[(p + lane) & mask]; /* one 256-byte row of the buffer: a dependent trip to memory */
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      a = __builtin_fma(a, b, 1.0); /* dependent double-precision chain */
      a = __builtin_fma(a, 0.999, b);
      m = fmaxf(m, __shfl_xor(m, 1 << (k % 6)));
      mine[(lane + k) & 1023] = __float_as_uint(m);
      __builtin_amdgcn_wave_barrier();
      const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine[k]); /* LDS -> scalar -> branch */
      if (s == 0x7fc00001u) b += 1.0;
    }
    p = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt) & mask;
  }
  if (lane == 0) out[blockIdx.x * 8 + wave] = a + m + p;
}

__global__ __launch_bounds__(256) void k_mfma(const uint32_t *w, int iters, float *out) {
  extern __shared__ uint32_t dyn[]; /* 96 KB */
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 24 * 1024; i += 256) dyn[i] = w[i & 2047];
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  u32x4 a[4];
  for (int t = 0; t < 4; ++t)
    for (int j = 0; j < 4; ++j) a[t][j] = w[(t * 4 + j) * 64 + lane];
  const uint32_t *base = dyn + wave * 6144 + lane * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const u32x4 b = *reinterpret_cast<const u32x4 *>(base + ((it + s) & 15) * 256);
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a[t]), "v"(b));
    }
  }
  float r = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < 16; ++i) r += acc[t][i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

static float run(hipStream_t ss, hipStream_t sm, int gs, int gm, const uint32_t *chase, uint32_t mask, const uint32_t *w, double *od,
                 float *of, int si, int mi) {
  hipEvent_t e0, e1, e2;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventCreate(&e2));
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, ss));
  CHECK(hipStreamWaitEvent(sm, e0, 0));
  if (gs) k_search<<<gs, 512, 0, ss>>>(chase, mask, si, od);
  if (gm) k_mfma<<<gm, 256, 96 * 1024, sm>>>(w, mi, of);
  CHECK(hipEventRecord(e1, sm));
  CHECK(hipStreamWaitEvent(ss, e1, 0));
  CHECK(hipEventRecord(e2, ss));
  CHECK(hipEventSynchronize(e2));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e2));
  return ms;
}

int main(int argc, char **argv) {
  int si = argc > 1 ? atoi(argv[1]) : 3000, mi = argc > 2 ? atoi(argv[2]) : 60000;
  const uint32_t n = 1u << 28; /* 1 GB of uint32_t: beyond the 256 MB Infinity Cache */
  uint32_t *chase, *w;
  double *od;
  float *of;
  CHECK(hipMalloc(&chase, (size_t)n * 4));
  {
    std::vector<uint32_t> h(1u << 22);
    uint32_t x = 12345u;
    for (auto &v : h) v = (x = x * 1664525u + 1013904223u);
    for (size_t off = 0; off < n; off += h.size()) CHECK(hipMemcpy(chase + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CHECK(hipMalloc(&w, 2048 * 4));
  {
    std::vector<uint32_t> h(2048, 0x3c003c00u); /* fp16 1.0 pairs */
    CHECK(hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CHECK(hipMalloc(&od, 8 * 4096 * 8));
  CHECK(hipMalloc(&of, 4096 * 256 * 4));
  CHECK(hipFuncSetAttribute((const void *)k_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  hipFuncAttributes fa;
  CHECK(hipFuncGetAttributes(&fa, (const void *)k_search));
  printf("k_search: %d registers, %zu B LDS, 8 wavefronts per workgroup\n", fa.numRegs, fa.sharedSizeBytes);
  CHECK(hipFuncGetAttributes(&fa, (const void *)k_mfma));
  printf("k_mfma:   %d registers, 96 KB LDS, 4 wavefronts per workgroup\n", fa.numRegs);
  for (int cus : {256, 64}) {
    hipStream_t ss, sm;
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < cus; ++c) m[(c * (256 / cus)) >> 5] |= 1u << ((c * (256 / cus)) & 31); /* every (256 / cus)-th unit */
    CHECK(hipExtStreamCreateWithCUMask(&ss, 8, m));
    CHECK(hipExtStreamCreateWithCUMask(&sm, 8, m));
    run(ss, sm, cus, cus, chase, n - 1, w, od, of, 10, 10); /* warm */
    for (int mult : {1, 2}) {
      const int g = cus * mult;
      const float ts = run(ss, sm, g, 0, chase, n - 1, w, od, of, si, mi);
      const float tm = run(ss, sm, 0, g, chase, n - 1, w, od, of, si, mi);
      const float tb = run(ss, sm, g, g, chase, n - 1, w, od, of, si, mi);
      printf("%3d compute units, %4d workgroups of each kind: search alone %.3f ms, MFMA alone %.3f ms, both at once %.3f ms "
             "= %.2f x the longer one, %.2f x the sum\n",
             cus, g, ts, tm, tb, tb / (ts > tm ? ts : tm), tb / (ts + tm));
    }
    CHECK(hipStreamDestroy(ss));
    CHECK(hipStreamDestroy(sm));
  }
  return 0;
}
