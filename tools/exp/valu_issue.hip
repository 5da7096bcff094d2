// Diagnostic microbenchmark: vector-instruction issue rate of ONE wave per SIMD against TWO, with and without
// an MFMA per slot (cycles from the shader clock).  usage: valu_issue [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int MF, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void kslot(const uint32_t *w, float *out, unsigned long long *cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  u32x4 a, b;
  for (int m = 0; m < 4; ++m) { a[m] = w[m * 64 + lane]; b[m] = w[(4 + m) * 64 + lane]; }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = (float)i;
  float zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (MF) {
        if (s & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[(s * NV + i) & 7]) : "v"(zero), "v"(f[(s * NV + i + 3) & 7]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV, int MF, int THREADS>
static void run(const uint32_t *dw, float *dout, unsigned long long *dc, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((kslot<NV, MF, THREADS>), dim3(256), dim3(THREADS), 0, 0, dw, dout, dc, iters / 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kslot<NV, MF, THREADS>), dim3(256), dim3(THREADS), 0, 0, dw, dout, dc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c = 0;
  hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("%d wave(s)/SIMD, %s, %2d vector instructions per slot: wave 0 %7.1f cycles per slot; kernel %6.2f ns per slot of every wave\n",
         THREADS / 256, MF ? "1 MFMA" : "no MFMA", NV, (double)c / ((double)iters * 8), ms * 1e6 / ((double)iters * 8));
}

int main(int argc, char **argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  uint32_t h[8 * 64];
  srand(1);
  for (auto &x : h) { uint32_t a = 0x3f00 + (rand() & 0xff), b = 0xbf00 + (rand() & 0xff); x = (a << 16) | b; }
  uint32_t *dw;
  float *dout;
  unsigned long long *dc;
  hipMalloc(&dw, sizeof h);
  hipMalloc(&dout, 256 * 512 * 4);
  hipMalloc(&dc, 8);
  hipMemcpy(dw, h, sizeof h, hipMemcpyHostToDevice);
#define ROW(MF, TH) run<0, MF, TH>(dw, dout, dc, iters); run<4, MF, TH>(dw, dout, dc, iters); run<8, MF, TH>(dw, dout, dc, iters); \
                    run<12, MF, TH>(dw, dout, dc, iters); run<16, MF, TH>(dw, dout, dc, iters);
  ROW(0, 256) ROW(1, 256) ROW(0, 512) ROW(1, 512)
  return 0;
}
