// Diagnostic microbenchmark: do packed / dot vector instructions overlap an MFMA of the same wave?  One wave per SIMD;
// slot = one v_mfma_f32_32x32x16_bf16 + 6 plain v_add_f32 + 2 instructions of the kind under test, placed right behind
// the MFMA (early) or behind the plain ones (late).  usage: mfma_pk [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void special(int i, f32x2 (&r)[8], float (&f)[8], uint32_t &pk, uint32_t sel, float &acc_a) {
  if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i & 7]) : "v"(r[(i + 4) & 7]));
  if (KIND == 1) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[i & 7]) : "v"(pk), "v"(sel));
  if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(f[i & 7]), "v"(f[(i + 1) & 7]));
  if (KIND == 3) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f[i & 7]) : "a"(acc_a));
  if (KIND == 4) asm volatile("v_sub_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f[i & 7]) : "v"(f[(i + 3) & 7]));
  if (KIND == 5) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[i & 7]) : "v"(f[(i + 3) & 7]), "v"(f[(i + 5) & 7]));
}

template <int KIND, int LATE>
__global__ __launch_bounds__(256, 1) void kslot(const uint32_t *w, float *out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  u32x4 a, b;
  for (int m = 0; m < 4; ++m) { a[m] = w[m * 64 + lane]; b[m] = w[(4 + m) * 64 + lane]; }
  f32x2 r[8];
  float f[8], g[8];
  for (int i = 0; i < 8; ++i) { r[i] = (f32x2){(float)i, 1.0f}; f[i] = (float)i; g[i] = 0.5f * i; }
  float acc_a;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc_a) : "v"(f[3]));
  uint32_t pk = w[lane], sel = 0x0000bf80u;
  asm volatile("v_mov_b32 %0, %0" : "+v"(sel));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
      if (!LATE) { special<KIND>(2 * s, r, f, pk, sel, acc_a); special<KIND>(2 * s + 1, r, f, pk, sel, acc_a); }
#pragma unroll
      for (int i = 0; i < 6; ++i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(g[(s * 6 + i) & 7]) : "v"(g[(s * 6 + i + 3) & 7]), "v"(g[(s * 6 + i + 5) & 7]));
      if (LATE) { special<KIND>(2 * s, r, f, pk, sel, acc_a); special<KIND>(2 * s + 1, r, f, pk, sel, acc_a); }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y + f[i] + g[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)pk;
}

template <int KIND, int LATE>
static void run(const uint32_t *dw, float *dout, int iters, const char *name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((kslot<KIND, LATE>), dim3(256), dim3(256), 0, 0, dw, dout, iters / 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kslot<KIND, LATE>), dim3(256), dim3(256), 0, 0, dw, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("MFMA + 6 v_add_f32 + 2 %-18s %s: %6.2f ns per slot\n", name, LATE ? "late " : "early", ms * 1e6 / ((double)iters * 8));
}

int main(int argc, char **argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  uint32_t h[8 * 64];
  srand(1);
  for (auto &x : h) { uint32_t a = 0x3f00 + (rand() & 0xff), b = 0xbf00 + (rand() & 0xff); x = (a << 16) | b; }
  uint32_t *dw;
  float *dout;
  hipMalloc(&dw, sizeof h);
  hipMalloc(&dout, 256 * 256 * 4);
  hipMemcpy(dw, h, sizeof h, hipMemcpyHostToDevice);
#define BOTH(K, N) run<K, 0>(dw, dout, iters, N); run<K, 1>(dw, dout, iters, N);
  BOTH(5, "v_add_f32") BOTH(0, "v_pk_add_f32") BOTH(1, "v_dot2c_f32_bf16") BOTH(2, "v_cvt_pk_bf16_f32") BOTH(3, "v_accvgpr_read_b32") BOTH(4, "v_sub_f32_dpp")
  return 0;
}
