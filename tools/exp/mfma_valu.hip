// Diagnostic microbenchmark: how much vector work one wave can issue beside its own MFMAs.  One wave per SIMD
// (256 threads per workgroup, one workgroup per CU), a loop of slots = one v_mfma_f32_32x32x16_bf16 (two accumulators,
// alternating) + NV vector instructions of one kind, everything asm volatile so that the order is the source order.
//   kind 0: v_pk_add_f32, independent registers     kind 1: v_add_f32, one dependent chain
//   kind 2: v_accvgpr_read_b32                       kind 3: v_cvt_pk_bf16_f32 / v_lshlrev / v_and / v_pk_add chain (the split)
//   kind 4: s_nop 1 + s_mov_b64 vcc + 2 x v_cndmask_b32_dpp (counted as 2)   kind 5: v_mov_b32 independent
// usage: mfma_valu [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void valu(int i, f32x2 (&r)[8], float (&f)[8], float &acc_a, uint32_t &pk, float zero) {
  if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i & 7]) : "v"(r[(i + 4) & 7]));
  if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(f[1]));
  if (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f[i & 7]) : "a"(acc_a));
  if (KIND == 3) {
    switch (i % 4) {
      case 0: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(f[0]), "v"(f[1])); break;
      case 1: asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(f[2]) : "v"(pk)); break;
      case 2: asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(f[3]) : "v"(pk)); break;
      default: asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[0]) : "v"(f[2])); break;
    }
  }
  if (KIND == 4) {
    if ((i & 1) == 0)
      asm volatile("s_nop 1\n\ts_mov_b64 vcc, %4\n\t"
                   "v_cndmask_b32_dpp %0, %2, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                   "v_cndmask_b32_dpp %1, %3, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                   : "=&v"(f[2]), "=&v"(f[3]) : "v"(f[0]), "v"(f[1]), "s"(0x5555555555555555ull), "v"(zero) : "vcc");
  }
  if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i & 7]) : "v"(zero));
}

template <int NV, int KIND>
__global__ __launch_bounds__(256, 1) void kslot(const uint32_t *w, float *out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  u32x4 a, b;
  for (int m = 0; m < 4; ++m) { a[m] = w[m * 64 + lane]; b[m] = w[(4 + m) * 64 + lane]; }
  f32x2 r[8];
  float f[8];
  for (int i = 0; i < 8; ++i) { r[i] = (f32x2){(float)i, 1.0f}; f[i] = (float)i; }
  float acc_a;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc_a) : "v"(f[3]));
  uint32_t pk = 0;
  float zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < NV; ++i) valu<KIND>(s * NV + i, r, f, acc_a, pk, zero);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)pk;
}

template <int NV, int KIND>
static void run(const uint32_t *dw, float *dout, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((kslot<NV, KIND>), dim3(256), dim3(256), 0, 0, dw, dout, iters / 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kslot<NV, KIND>), dim3(256), dim3(256), 0, 0, dw, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("kind %d  %2d vector instructions per MFMA: %6.2f ns per slot\n", KIND, NV, ms * 1e6 / ((double)iters * 8));
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
#define ROW(K) run<0, K>(dw, dout, iters); run<2, K>(dw, dout, iters); run<4, K>(dw, dout, iters); run<6, K>(dw, dout, iters); \
               run<8, K>(dw, dout, iters); run<12, K>(dw, dout, iters); run<16, K>(dw, dout, iters);
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5)
  return 0;
}
