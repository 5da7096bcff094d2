// Diagnostic microbenchmark: two waves per SIMD with split roles (a producer / consumer form of a transform-heavy kernel).
// Waves 0..3 of a 512-thread workgroup issue the MFMAs: one v_mfma_f32_32x32x16_bf16 per slot and, per step of 12 slots,
// NA weight-fragment reads + 3 operand reads (ds_read_b128, used a step later).  Waves 4..7 (same SIMDs) issue NV plain
// vector instructions per slot and 3 ds_write_b128 per step.  BAR: an s_barrier closes every step.
// usage: prodcons [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NA, int BAR, int KIND>
__global__ __launch_bounds__(512, 1) void kpc(const uint32_t *w, float *out, int iters) {
  extern __shared__ u32x4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 12 * 64; i += 512) lds[i] = (u32x4){w[i & 511], w[(i + 1) & 511], w[(i + 2) & 511], w[(i + 3) & 511]};
  __syncthreads();
  float s = 0;
  if (wave < 4) {
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    u32x4 a[2][6], b[2][3];
    for (int q = 0; q < 2; ++q) {
      for (int f = 0; f < 6; ++f) a[q][f] = lds[(f * 64 + lane)];
      for (int f = 0; f < 3; ++f) b[q][f] = lds[((6 + f) * 64 + lane)];
    }
    const u32x4 *mine = lds + (wave & 3) * 12 * 64 + lane;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int q = 0; q < 2; ++q) { /* step parity: use set q, load set q ^ 1 */
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          if (k & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a[q][k % 6]), "v"(b[q][k % 3]));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a[q][k % 6]), "v"(b[q][k % 3]));
          if (k < NA) a[q ^ 1][k] = mine[k * 64];
          if (k >= 6 && k < 9) b[q ^ 1][k - 6] = mine[(6 + k - 6) * 64];
        }
        if (BAR) __builtin_amdgcn_s_barrier();
      }
    }
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  } else {
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)i + lane;
    float acc_a;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc_a) : "v"(f[3]));
    u32x4 *mine = lds + (wave & 3) * 12 * 64 + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int q = k * NV + i;
          if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[q & 7]) : "v"(f[(q + 3) & 7]), "v"(f[(q + 5) & 7]));
          else if (q % 4 == 0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f[q & 7]) : "a"(acc_a));
          else if (q % 4 == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(f[q & 7]) : "v"(f[(q + 3) & 7]), "v"(f[(q + 5) & 7]));
          else asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[q & 7]) : "v"(f[(q + 3) & 7]), "v"(f[(q + 5) & 7]));
        }
        if (k % 4 == 3) mine[(6 + k / 4) * 64] = (u32x4){__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
      }
      if (BAR) __builtin_amdgcn_s_barrier();
    }
    for (int i = 0; i < 8; ++i) s += f[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV, int NA, int BAR, int KIND>
static void run(const uint32_t *dw, float *dout, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((kpc<NV, NA, BAR, KIND>), dim3(256), dim3(512), 4 * 12 * 64 * 16, 0, dw, dout, iters / 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kpc<NV, NA, BAR, KIND>), dim3(256), dim3(512), 4 * 12 * 64 * 16, 0, dw, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("MFMA waves: %d fragment + 3 operand reads per step; vector waves: %2d %s instructions per slot + 3 writes per step; barrier per step %s: %6.2f ns per slot\n",
         NA, NV, KIND ? "mixed" : "v_add_f32", BAR ? "yes" : "no ", ms * 1e6 / ((double)iters * 12));
}

int main(int argc, char **argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 10000;
  uint32_t h[512];
  srand(1);
  for (auto &x : h) { uint32_t a = 0x3f00 + (rand() & 0xff), b = 0xbf00 + (rand() & 0xff); x = (a << 16) | b; }
  uint32_t *dw;
  float *dout;
  hipMalloc(&dw, sizeof h);
  hipMalloc(&dout, 256 * 512 * 4);
  hipMemcpy(dw, h, sizeof h, hipMemcpyHostToDevice);
  run<0, 0, 0, 0>(dw, dout, iters); run<0, 6, 0, 0>(dw, dout, iters); run<0, 6, 1, 0>(dw, dout, iters);
  run<4, 6, 1, 0>(dw, dout, iters); run<6, 6, 1, 0>(dw, dout, iters); run<7, 6, 1, 0>(dw, dout, iters); run<8, 6, 1, 0>(dw, dout, iters);
  run<6, 6, 1, 1>(dw, dout, iters); run<7, 6, 1, 1>(dw, dout, iters); run<8, 6, 1, 1>(dw, dout, iters);
  return 0;
}
