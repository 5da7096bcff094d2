// Diagnostic microbenchmark: two waves per SIMD with split roles.  Waves 0..3 of a 512-thread workgroup issue the MFMAs
// (one v_mfma_f32_32x32x16_bf16 per slot, a ds_read_b128 every second slot), waves 4..7 (same SIMDs) issue NV vector
// instructions per slot (kind 0: v_add_f32, kind 1: v_pk_add_f32, kind 2: v_dot2c_f32_bf16) and three ds_write_b128 per
// 12 slots; an s_barrier closes every 12 slots.  usage: prodcons [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND, int BARRIER>
__global__ __launch_bounds__(512, 1) void kpc(const uint32_t *w, float *out, int iters) {
  __shared__ u32x4 xch[8 * 64 * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  u32x4 a, b, b2;
  for (int m = 0; m < 4; ++m) { a[m] = w[m * 64 + lane]; b[m] = w[(4 + m) * 64 + lane]; b2[m] = b[m]; }
  f32x2 r[8];
  float f[8];
  for (int i = 0; i < 8; ++i) { r[i] = (f32x2){(float)i, 1.0f}; f[i] = (float)i; }
  uint32_t pk = w[lane], sel = 0x0000bf80u;
  asm volatile("v_mov_b32 %0, %0" : "+v"(sel));
  u32x4 *mine = xch + (wave & 3) * 64 * 3 + lane;
  for (int it = 0; it < iters; ++it) {
    if (wave < 4) {
#pragma unroll
      for (int s = 0; s < 12; ++s) {
        /* the operand read in slot s is used from slot s + 6 on (two register sets) */
        u32x4 &bb = (s / 6) & 1 ? b2 : b;
        if (s & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(bb));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(bb));
        if (s == 0) b2 = mine[0];
        if (s == 6) b = mine[64];
      }
    } else {
#pragma unroll
      for (int s = 0; s < 12; ++s) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int q = s * NV + i;
          if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[q & 7]) : "v"(f[(q + 3) & 7]), "v"(f[(q + 5) & 7]));
          if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[q & 7]) : "v"(r[(q + 4) & 7]));
          if (KIND == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[q & 7]) : "v"(pk), "v"(sel));
        }
        if (s % 4 == 3) { u32x4 t = {__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(r[0].x), pk}; mine[(s / 4) * 64] = t; }
      }
    }
    if (BARRIER) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y + f[i];
  out[blockIdx.x * 512 + threadIdx.x] = s + (float)b[0] + (float)b2[1];
}

template <int NV, int KIND, int BARRIER>
static void run(const uint32_t *dw, float *dout, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((kpc<NV, KIND, BARRIER>), dim3(256), dim3(512), 0, 0, dw, dout, iters / 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kpc<NV, KIND, BARRIER>), dim3(256), dim3(512), 0, 0, dw, dout, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const char *names[3] = {"v_add_f32", "v_pk_add_f32", "v_dot2c_f32_bf16"};
  printf("MFMA wave + vector wave, %2d x %-17s per slot, barrier every 12 slots %s: %6.2f ns per slot\n", NV, names[KIND],
         BARRIER ? "yes" : "no ", ms * 1e6 / ((double)iters * 12));
}

int main(int argc, char **argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 10000;
  uint32_t h[8 * 64];
  srand(1);
  for (auto &x : h) { uint32_t a = 0x3f00 + (rand() & 0xff), b = 0xbf00 + (rand() & 0xff); x = (a << 16) | b; }
  uint32_t *dw;
  float *dout;
  hipMalloc(&dw, sizeof h);
  hipMalloc(&dout, 256 * 512 * 4);
  hipMemcpy(dw, h, sizeof h, hipMemcpyHostToDevice);
  run<0, 0, 0>(dw, dout, iters); run<0, 0, 1>(dw, dout, iters);
  run<4, 0, 1>(dw, dout, iters); run<6, 0, 1>(dw, dout, iters); run<7, 0, 1>(dw, dout, iters); run<8, 0, 1>(dw, dout, iters); run<10, 0, 1>(dw, dout, iters);
  run<4, 1, 1>(dw, dout, iters); run<6, 1, 1>(dw, dout, iters); run<8, 1, 1>(dw, dout, iters);
  run<4, 2, 1>(dw, dout, iters); run<6, 2, 1>(dw, dout, iters); run<8, 2, 1>(dw, dout, iters);
  return 0;
}
