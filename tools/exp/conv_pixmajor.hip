// Experiment (round 4): a padding-free formulation of rescnn4's 3x3 convolutions on the 4x4 board.
//
// The product kernel (csrc/nn_rescnn.hip) makes an MFMA column a (position, pixel) pair: a tap is a DPP row shift of
// the activation registers, and the 44 of 144 (pixel, tap) pairs that fall outside the board multiply zeros -- 31 % of the
// matrix work.  Here an MFMA column is a POSITION (32 per workgroup) and every output pixel has its own accumulators:
//     D_p[co, pos] += W_tap[co, ci] . X_q[ci, pos]      for the taps whose source pixel q = p + tap lies on the board
// so only the 100 real (pixel, tap) pairs are multiplied.  The price: the neighbour pixel's activations are another wave's
// registers, so activations travel through LDS -- 16 pixels x 64 channels x 32 positions x two fp16 terms = 128 KB per
// workgroup, written by the epilogue of every convolution and read back as B fragments (one ds_read_b128 per MFMA
// instead of one per three), with one more barrier pair per convolution; weights stream one tap (16 KB) at a time through
// the remaining 32 KB.  Eight waves: waves 0-3 own an interior pixel (9 taps) and a corner (4), waves 4-7 two edge pixels
// of one side (6 + 6): 13 / 12 tap-pixels per wave, 300 MFMAs per wave and convolution on average against 432.
//
// This file times EIGHT 64 -> 64 convolutions (the trunk of rescnn4 without stem and heads) with the f16x3 arithmetic of the
// product (x = fp16(x) + fp16(x - fp16(x)), products w0 x0 + w0 x1 + w1 x0, fp32 accumulate, ReLU epilogue) and checks
// the first workgroup against a float64 evaluation on the host.
//   hipcc --offload-arch=gfx950 -O3 -I corintho_ai_amd/csrc tools/exp/conv_pixmajor.hip -o build_ab/conv_pixmajor && build_ab/conv_pixmajor [rows]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "lds_dma.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define NCONV 8
#define X_WORDS (16 * 4 * 2 * 256)  /* pixel x K step x term fragments of 1 KiB: 128 KB */
#define TAP_WORDS (4 * 2 * 2 * 256) /* K step x output tile x term fragments: 16 KB */
#define LDS_WORDS (X_WORDS + 2 * TAP_WORDS)

__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void split2(float a, float b, uint32_t &hi, uint32_t &lo) {
  f32x2 v = {a, b};
  f16x2 h = __builtin_convertvector(v, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  f32x2 hf = __builtin_convertvector(h, f32x2);
  f32x2 r = {v.x - hf.x, v.y - hf.y};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}

/* xin: [workgroup][X_WORDS] activations in fragment order; w: [conv][tap][TAP_WORDS]; out: [workgroup][16 px][64 co][32 pos] */
__global__ __launch_bounds__(512, 2) void conv_pixmajor(const uint32_t *__restrict__ xin, const uint32_t *__restrict__ w, float *__restrict__ out,
                                                       int nwg, unsigned long long *stamps) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  if ((int)blockIdx.x >= nwg) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds_base = co_lds_addr(lds);
  uint32_t *X = lds;
  const uint32_t *Wb = lds + X_WORDS;
  /* this wave's two output pixels */
  const int P0 = wave < 4 ? (wave == 0 ? 5 : wave == 1 ? 6 : wave == 2 ? 9 : 10) : (wave == 4 ? 1 : wave == 5 ? 4 : wave == 6 ? 7 : 13);
  const int P1 = wave < 4 ? (wave == 0 ? 0 : wave == 1 ? 3 : wave == 2 ? 12 : 15) : (wave == 4 ? 2 : wave == 5 ? 8 : wave == 6 ? 11 : 14);
  int valid[2] = {0, 0};
  for (int pi = 0; pi < 2; ++pi) {
    const int p = pi ? P1 : P0, y = p >> 2, x = p & 3;
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      if (y + dy >= 0 && y + dy < 4 && x + dx >= 0 && x + dx < 4) valid[pi] |= 1 << tap;
    }
  }
  /* activations of the 32 positions: 128 pieces of 1 KiB, 16 per wave; then tap 0 of convolution 0 */
  const uint32_t *xg = xin + (size_t)blockIdx.x * X_WORDS + lane * 4;
#pragma unroll
  for (int i = 0; i < 16; ++i) co_lds_dma_1k(xg + (wave * 16 + i) * 256, lds_base + (uint32_t)(wave * 16 + i) * 1024u);
  auto stage_tap = [&](int g) { /* global tap counter g = conv * 9 + tap: 16 pieces, two per wave */
    const uint32_t *src = w + (size_t)g * TAP_WORDS + lane * 4;
    const uint32_t dst = lds_base + (uint32_t)(X_WORDS + (g & 1) * TAP_WORDS) * 4u;
    co_lds_dma_1k(src + (2 * wave) * 256, dst + (uint32_t)(2 * wave) * 1024u);
    co_lds_dma_1k(src + (2 * wave + 1) * 256, dst + (uint32_t)(2 * wave + 1) * 1024u);
  };
  stage_tap(0);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  f32x16 acc[2][2];
  int g = 0;
  for (int conv = 0; conv < NCONV; ++conv) {
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[pi][to][i] = 0.0f;
    for (int tap = 0; tap < 9; ++tap, ++g) {
      CO_WAIT_VMCNT(0); /* this wave's pieces of tap g have landed (requested one tap ago) */
      co_wg_barrier();  /* ... every wave's; everyone has left the other buffer (and, for tap 0, has written its activations) */
      if (g + 1 < NCONV * 9) stage_tap(g + 1);
      const uint32_t *wb = Wb + (g & 1) * TAP_WORDS + lane * 4;
      const int dq = (tap / 3 - 1) * 4 + (tap % 3 - 1);
      const bool v0 = (valid[0] >> tap) & 1, v1 = (valid[1] >> tap) & 1;
      if (!v0 && !v1) continue;
      const uint32_t *x0 = X + ((P0 + dq) * 4 * 2) * 256 + lane * 4, *x1 = X + ((P1 + dq) * 4 * 2) * 256 + lane * 4;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        u32x4 a[2][2];
#pragma unroll
        for (int to = 0; to < 2; ++to)
#pragma unroll
          for (int t = 0; t < 2; ++t) a[t][to] = *reinterpret_cast<const u32x4 *>(wb + ((s * 2 + to) * 2 + t) * 256);
        if (v0) {
          const u32x4 b0 = *reinterpret_cast<const u32x4 *>(x0 + (s * 2 + 0) * 256), b1 = *reinterpret_cast<const u32x4 *>(x0 + (s * 2 + 1) * 256);
#pragma unroll
          for (int to = 0; to < 2; ++to) acc[0][to] = mfma(a[0][to], b0, acc[0][to]);
#pragma unroll
          for (int to = 0; to < 2; ++to) acc[0][to] = mfma(a[0][to], b1, acc[0][to]);
#pragma unroll
          for (int to = 0; to < 2; ++to) acc[0][to] = mfma(a[1][to], b0, acc[0][to]);
        }
        if (v1) {
          const u32x4 b0 = *reinterpret_cast<const u32x4 *>(x1 + (s * 2 + 0) * 256), b1 = *reinterpret_cast<const u32x4 *>(x1 + (s * 2 + 1) * 256);
#pragma unroll
          for (int to = 0; to < 2; ++to) acc[1][to] = mfma(a[0][to], b0, acc[1][to]);
#pragma unroll
          for (int to = 0; to < 2; ++to) acc[1][to] = mfma(a[0][to], b1, acc[1][to]);
#pragma unroll
          for (int to = 0; to < 2; ++to) acc[1][to] = mfma(a[1][to], b0, acc[1][to]);
        }
      }
    }
    /* epilogue: ReLU, two fp16 terms, back to LDS as the B fragments of the next convolution.  Register 8a + j of tile
     * `to` is k-slot j of step s = 2 to + a (the accumulator layout is the operand layout, as in the product kernel). */
    co_wg_barrier(); /* every wave has read the last activations of this convolution */
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int p = pi ? P1 : P0;
#pragma unroll
      for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
          u32x4 hi, lo;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            float v0f = acc[pi][to][8 * a2 + 2 * m], v1f = acc[pi][to][8 * a2 + 2 * m + 1];
            v0f = v0f > 0.0f ? v0f : 0.0f;
            v1f = v1f > 0.0f ? v1f : 0.0f;
            uint32_t h, l;
            split2(v0f, v1f, h, l);
            hi[m] = h;
            lo[m] = l;
          }
          const int s = 2 * to + a2;
          *reinterpret_cast<u32x4 *>(X + ((p * 4 + s) * 2 + 0) * 256 + lane * 4) = hi;
          *reinterpret_cast<u32x4 *>(X + ((p * 4 + s) * 2 + 1) * 256 + lane * 4) = lo;
        }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && stamps) stamps[blockIdx.x] = c1 - c0;
  /* the last activations (before the split) of this wave's pixels: out[wg][px][co][pos] */
  if (out) {
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int p = pi ? P1 : P0;
#pragma unroll
      for (int to = 0; to < 2; ++to)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = 32 * to + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
          float v = acc[pi][to][r];
          out[(((size_t)blockIdx.x * 16 + p) * 64 + co) * 32 + (lane & 31)] = v > 0.0f ? v : 0.0f;
        }
    }
  }
}

static uint16_t f16_bits(float f) {
  _Float16 h = (_Float16)f;
  uint16_t u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}
static float f16_val(uint16_t u) {
  _Float16 h;
  __builtin_memcpy(&h, &u, 2);
  return (float)h;
}

int main(int argc, char **argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 32768;
  const int nwg = (rows + 31) / 32;
  srand(3);
  auto rnd = []() { return (float)rand() / (float)RAND_MAX; };
  /* weights [conv][tap][ci][co], activations [wg][pos][px][ci] */
  std::vector<float> W((size_t)NCONV * 9 * 64 * 64), X0((size_t)nwg * 32 * 16 * 64);
  for (auto &v : W) v = 0.14f * (rnd() - 0.5f);
  for (auto &v : X0) v = rnd();
  /* fragment packing.  k-slot (h, j) of step s = 2T + a <-> channel 32T + 8(2a + j/4) + 4h + j%4 */
  auto chan = [](int s, int h, int j) { return 32 * (s >> 1) + 8 * (2 * (s & 1) + (j >> 2)) + 4 * h + (j & 3); };
  std::vector<uint32_t> hw((size_t)NCONV * 9 * TAP_WORDS, 0u), hx((size_t)nwg * X_WORDS, 0u);
  for (int c = 0; c < NCONV; ++c)
    for (int tap = 0; tap < 9; ++tap)
      for (int s = 0; s < 4; ++s)
        for (int to = 0; to < 2; ++to)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
              const int co = 32 * to + (l & 31), ci = chan(s, l >> 5, j);
              float v = W[(((size_t)c * 9 + tap) * 64 + ci) * 64 + co];
              const uint16_t h0 = f16_bits(v), h1 = f16_bits(v - f16_val(h0));
              const size_t base = ((size_t)c * 9 + tap) * TAP_WORDS;
              hw[base + (((s * 2 + to) * 2 + 0) * 64 + l) * 4 + j / 2] |= (uint32_t)h0 << (16 * (j & 1));
              hw[base + (((s * 2 + to) * 2 + 1) * 64 + l) * 4 + j / 2] |= (uint32_t)h1 << (16 * (j & 1));
            }
  for (int wg = 0; wg < nwg; ++wg)
    for (int q = 0; q < 16; ++q)
      for (int s = 0; s < 4; ++s)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) {
            const int pos = l & 31, ci = chan(s, l >> 5, j);
            float v = X0[(((size_t)wg * 32 + pos) * 16 + q) * 64 + ci];
            const uint16_t h0 = f16_bits(v), h1 = f16_bits(v - f16_val(h0));
            const size_t base = (size_t)wg * X_WORDS;
            hx[base + (((q * 4 + s) * 2 + 0) * 64 + l) * 4 + j / 2] |= (uint32_t)h0 << (16 * (j & 1));
            hx[base + (((q * 4 + s) * 2 + 1) * 64 + l) * 4 + j / 2] |= (uint32_t)h1 << (16 * (j & 1));
          }
  uint32_t *dw, *dx;
  float *dout;
  unsigned long long *ds;
  hipMalloc(&dw, hw.size() * 4);
  hipMalloc(&dx, hx.size() * 4);
  hipMalloc(&dout, (size_t)nwg * 16 * 64 * 32 * 4);
  hipMalloc(&ds, (size_t)nwg * 8);
  hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  const size_t lds_bytes = (size_t)LDS_WORDS * 4;
  if (hipFuncSetAttribute((const void *)conv_pixmajor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
    printf("cannot reserve %zu bytes of LDS\n", lds_bytes);
    return 1;
  }
  hipLaunchKernelGGL(conv_pixmajor, dim3(nwg), dim3(512), lds_bytes, 0, dx, dw, dout, nwg, ds);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
    return 1;
  }
  /* ---- check workgroup 0 against float64 */
  {
    std::vector<double> cur((size_t)32 * 16 * 64), nxt(cur.size());
    for (int pos = 0; pos < 32; ++pos)
      for (int q = 0; q < 16; ++q)
        for (int ci = 0; ci < 64; ++ci) cur[((size_t)pos * 16 + q) * 64 + ci] = X0[(((size_t)0 * 32 + pos) * 16 + q) * 64 + ci];
    for (int c = 0; c < NCONV; ++c) {
      for (int pos = 0; pos < 32; ++pos)
        for (int p = 0; p < 16; ++p)
          for (int co = 0; co < 64; ++co) {
            double acc = 0.0;
            for (int tap = 0; tap < 9; ++tap) {
              const int y = (p >> 2) + tap / 3 - 1, x = (p & 3) + tap % 3 - 1;
              if (y < 0 || y > 3 || x < 0 || x > 3) continue;
              const double *xr = &cur[((size_t)pos * 16 + y * 4 + x) * 64];
              const float *wr = &W[(((size_t)c * 9 + tap) * 64) * 64 + co];
              for (int ci = 0; ci < 64; ++ci) acc += xr[ci] * (double)wr[(size_t)ci * 64];
            }
            nxt[((size_t)pos * 16 + p) * 64 + co] = acc > 0.0 ? acc : 0.0;
          }
      cur.swap(nxt);
    }
    std::vector<float> got((size_t)16 * 64 * 32);
    hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0.0, scale = 0.0;
    for (int pos = 0; pos < 32; ++pos)
      for (int p = 0; p < 16; ++p)
        for (int co = 0; co < 64; ++co) {
          const double want = cur[((size_t)pos * 16 + p) * 64 + co], have = got[((size_t)p * 64 + co) * 32 + pos];
          worst = std::max(worst, fabs(want - have));
          scale = std::max(scale, fabs(want));
        }
    printf("check against float64 after %d convolutions: max abs error %.3g (largest activation %.3g) -- %s\n", NCONV, worst, scale,
           worst <= 2e-5 * std::max(scale, 1.0) ? "OK" : "MISMATCH");
    if (!(worst <= 2e-5 * std::max(scale, 1.0))) return 2;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 2000;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv_pixmajor, dim3(nwg), dim3(512), lds_bytes, 0, dx, dw, (float *)nullptr, nwg, ds);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv_pixmajor, dim3(nwg), dim3(512), lds_bytes, 0, dx, dw, (float *)nullptr, nwg, ds);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hs(nwg);
  hipMemcpy(hs.data(), ds, (size_t)nwg * 8, hipMemcpyDeviceToHost);
  std::sort(hs.begin(), hs.end());
  const double flop_useful = (double)rows * NCONV * 100.0 * 64 * 64 * 2, flop_padded = flop_useful * 1.44;
  printf("pixel-major, %d rows (%d workgroups): %.4f ms per launch of %d convolutions; %.0f core cycles per workgroup pass (median); "
         "%.1f TFLOP/s of real products (x3 issued), %.1f TFLOP/s as the product kernel counts them (with padding)\n",
         rows, nwg, ms / reps, NCONV, (double)hs[nwg / 2], flop_useful * reps / (ms * 1e-3) / 1e12, flop_padded * reps / (ms * 1e-3) / 1e12);
  return 0;
}
