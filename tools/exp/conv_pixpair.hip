// Experiment (round 4, after K6p's tap analysis): the PIXEL-PAIR column formulation of rescnn4's 3x3 convolutions.
//
// K6p (csrc/nn_rescnn.hip, tools/exp/conv_pixmajor.hip) makes an MFMA column a position, 32 per workgroup, and needs all
// 160 KB of LDS: one workgroup per CU, whose tap barriers, per-tap imbalance and nine epilogues leave the MFMA pipe idle
// 38 % of a pass.  Here a workgroup has 16 positions and a column is (one of a PAIR of output pixels, position): the two
// pixels of a pair use the same tap's weights, so one 32x32x16 serves both.  Pairs: the interior pixels (5,6) (9,10): 9
// taps; the edge pixels of one side (1,2) (13,14) (4,8) (7,11): the same 6 taps; the corners (0,3) (12,15): 6 taps in the
// union, in 4 of which one half of the columns is zeroed -- 54 pair-slots per convolution where the board has 50.
// Per accumulator the products and their order are K6p's (bit-identical).  LDS: 64 KB of activations + two half-tap
// weight slots of 8 KB = 80 KB, 128 registers per wave: TWO workgroups per CU, each one's barriers and epilogue under the
// other's MFMAs.  The price: the weight stream per position doubles, a barrier per half tap.
//
// This file times EIGHT 64 -> 64 convolutions like conv_pixmajor.hip and checks workgroup 0 against float64.
//   hipcc --offload-arch=gfx950 -O3 -I corintho_ai_amd/csrc tools/exp/conv_pixpair.hip -o build_ab/conv_pixpair && build_ab/conv_pixpair [rows]
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
#define NPOS 16
#define X_WORDS (16 * 4 * 2 * 128)   /* pixel x K step x term x (k half x position) fragments of 512 B: 64 KB */
#define TAP_WORDS (4 * 2 * 2 * 256)  /* K step x output tile x term fragments: 16 KB, staged in two halves */
#define HALF_WORDS (TAP_WORDS / 2)
#define LDS_WORDS (X_WORDS + 2 * HALF_WORDS)

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

/* xin: [workgroup][X_WORDS] activations in fragment order; w: [conv][tap][TAP_WORDS]; out: [workgroup][16 px][64 co][16 pos] */
__global__ __launch_bounds__(512, 4) void conv_pixpair(const uint32_t *__restrict__ xin, const uint32_t *__restrict__ w, float *__restrict__ out,
                                                      int nwg, unsigned long long *stamps) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  if ((int)blockIdx.x >= nwg) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, sel = n >> 4, pos = n & 15, h = lane >> 5;
  const uint32_t lds_base = co_lds_addr(lds);
  uint32_t *X = lds;
  const uint32_t *Wb = lds + X_WORDS;
  /* this wave's pair of output pixels: waves w and w + 4 share a SIMD -- an interior pair with a corner pair, edges with edges */
  const int PA = wave == 0 ? 5 : wave == 1 ? 9 : wave == 2 ? 1 : wave == 3 ? 4 : wave == 4 ? 0 : wave == 5 ? 12 : wave == 6 ? 13 : 7;
  const int PB = wave == 0 ? 6 : wave == 1 ? 10 : wave == 2 ? 2 : wave == 3 ? 8 : wave == 4 ? 3 : wave == 5 ? 15 : wave == 6 ? 14 : 11;
  int va = 0, vb = 0;
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    if ((PA >> 2) + dy >= 0 && (PA >> 2) + dy < 4 && (PA & 3) + dx >= 0 && (PA & 3) + dx < 4) va |= 1 << tap;
    if ((PB >> 2) + dy >= 0 && (PB >> 2) + dy < 4 && (PB & 3) + dx >= 0 && (PB & 3) + dx < 4) vb |= 1 << tap;
  }
  const int myq = sel ? PB : PA;       /* this lane's output pixel */
  const int myv = sel ? vb : va;       /* its taps on the board */
  const uint32_t *xg = xin + (size_t)blockIdx.x * X_WORDS + lane * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) co_lds_dma_1k(xg + (wave * 8 + i) * 256, lds_base + (uint32_t)(wave * 8 + i) * 1024u);
  auto stage_unit = [&](int u) { /* half tap u = (conv * 9 + tap) * 2 + half: 8 pieces, one per wave */
    const uint32_t *src = w + (size_t)u * HALF_WORDS + lane * 4;
    const uint32_t dst = lds_base + (uint32_t)(X_WORDS + (u & 1) * HALF_WORDS) * 4u;
    co_lds_dma_1k(src + wave * 256, dst + (uint32_t)wave * 1024u);
  };
  stage_unit(0);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  f32x16 acc[2];
  int u = 0;
  for (int conv = 0; conv < NCONV; ++conv) {
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[to][i] = 0.0f;
    for (int tap = 0; tap < 9; ++tap) {
      const bool active = ((va | vb) >> tap) & 1, both = ((va & vb) >> tap) & 1;
      const int dq = (tap / 3 - 1) * 4 + (tap % 3 - 1);
      const bool lane_on = (myv >> tap) & 1;
      const int q = lane_on ? myq + dq : myq; /* (an off-board lane reads its own pixel and is zeroed) */
      const uint32_t *xq = X + ((q * 4 * 2 * 2 + h) * 16 + pos) * 4;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh, ++u) {
        CO_WAIT_VMCNT(0); /* this wave's piece of unit u has landed (requested one unit ago) */
        co_wg_barrier();  /* ... every wave's; everyone has left the other slot (and, at a convolution's start, written its activations) */
        if (u + 1 < NCONV * 18) stage_unit(u + 1);
        if (!active) continue;
        const uint32_t *wb = Wb + (u & 1) * HALF_WORDS + lane * 4;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int s = 2 * hh + s2;
          u32x4 a[2][2];
#pragma unroll
          for (int to = 0; to < 2; ++to)
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t][to] = *reinterpret_cast<const u32x4 *>(wb + ((s2 * 2 + to) * 2 + t) * 256);
          u32x4 b0 = *reinterpret_cast<const u32x4 *>(xq + ((s * 2 + 0) * 2) * 64), b1 = *reinterpret_cast<const u32x4 *>(xq + ((s * 2 + 1) * 2) * 64);
          if (!both) { /* a corner pair at a tap only one of its pixels has: the other pixel's columns multiply zeros */
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              b0[m] = lane_on ? b0[m] : 0u;
              b1[m] = lane_on ? b1[m] : 0u;
            }
          }
#pragma unroll
          for (int to = 0; to < 2; ++to) acc[to] = mfma(a[0][to], b0, acc[to]);
#pragma unroll
          for (int to = 0; to < 2; ++to) acc[to] = mfma(a[0][to], b1, acc[to]);
#pragma unroll
          for (int to = 0; to < 2; ++to) acc[to] = mfma(a[1][to], b0, acc[to]);
        }
      }
    }
    /* epilogue: ReLU, two fp16 terms, back to LDS as the B fragments of the next convolution (register 8a + j of tile `to` is
     * k-slot j of step 2 to + a, for this lane's pixel, position and k half) */
    co_wg_barrier(); /* every wave has read the last activations of this convolution */
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2) {
        u32x4 hi, lo;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          float v0f = acc[to][8 * a2 + 2 * m], v1f = acc[to][8 * a2 + 2 * m + 1];
          v0f = v0f > 0.0f ? v0f : 0.0f;
          v1f = v1f > 0.0f ? v1f : 0.0f;
          uint32_t hw_, lw_;
          split2(v0f, v1f, hw_, lw_);
          hi[m] = hw_;
          lo[m] = lw_;
        }
        const int s = 2 * to + a2;
        *reinterpret_cast<u32x4 *>(X + ((((myq * 4 + s) * 2 + 0) * 2 + h) * 16 + pos) * 4) = hi;
        *reinterpret_cast<u32x4 *>(X + ((((myq * 4 + s) * 2 + 1) * 2 + h) * 16 + pos) * 4) = lo;
      }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && stamps) stamps[blockIdx.x] = c1 - c0;
  if (out) {
#pragma unroll
    for (int to = 0; to < 2; ++to)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * to + 8 * (r >> 2) + 4 * h + (r & 3);
        float v = acc[to][r];
        out[(((size_t)blockIdx.x * 16 + myq) * 64 + co) * NPOS + pos] = v > 0.0f ? v : 0.0f;
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
  const int nwg = (rows + NPOS - 1) / NPOS;
  srand(3);
  auto rnd = []() { return (float)rand() / (float)RAND_MAX; };
  /* weights [conv][tap][ci][co], activations [wg][pos][px][ci] */
  std::vector<float> W((size_t)NCONV * 9 * 64 * 64), X0((size_t)nwg * NPOS * 16 * 64);
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
        for (int hh = 0; hh < 2; ++hh)
          for (int pos = 0; pos < NPOS; ++pos)
            for (int j = 0; j < 8; ++j) {
              const int ci = chan(s, hh, j);
              float v = X0[(((size_t)wg * NPOS + pos) * 16 + q) * 64 + ci];
              const uint16_t h0 = f16_bits(v), h1 = f16_bits(v - f16_val(h0));
              const size_t base = (size_t)wg * X_WORDS;
              hx[base + ((((q * 4 + s) * 2 + 0) * 2 + hh) * 16 + pos) * 4 + j / 2] |= (uint32_t)h0 << (16 * (j & 1));
              hx[base + ((((q * 4 + s) * 2 + 1) * 2 + hh) * 16 + pos) * 4 + j / 2] |= (uint32_t)h1 << (16 * (j & 1));
            }
  uint32_t *dw, *dx;
  float *dout;
  unsigned long long *ds;
  hipMalloc(&dw, hw.size() * 4);
  hipMalloc(&dx, hx.size() * 4);
  hipMalloc(&dout, (size_t)nwg * 16 * 64 * NPOS * 4);
  hipMalloc(&ds, (size_t)nwg * 8);
  hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  const size_t lds_bytes = (size_t)LDS_WORDS * 4;
  if (hipFuncSetAttribute((const void *)conv_pixpair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
    printf("cannot reserve %zu bytes of LDS\n", lds_bytes);
    return 1;
  }
  hipLaunchKernelGGL(conv_pixpair, dim3(nwg), dim3(512), lds_bytes, 0, dx, dw, dout, nwg, ds);
  if (hipDeviceSynchronize() != hipSuccess) {
    printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
    return 1;
  }
  /* ---- check workgroup 0 against float64 */
  {
    std::vector<double> cur((size_t)NPOS * 16 * 64), nxt(cur.size());
    for (int pos = 0; pos < NPOS; ++pos)
      for (int q = 0; q < 16; ++q)
        for (int ci = 0; ci < 64; ++ci) cur[((size_t)pos * 16 + q) * 64 + ci] = X0[(((size_t)0 * NPOS + pos) * 16 + q) * 64 + ci];
    for (int c = 0; c < NCONV; ++c) {
      for (int pos = 0; pos < NPOS; ++pos)
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
    std::vector<float> got((size_t)16 * 64 * NPOS);
    hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0.0, scale = 0.0;
    for (int pos = 0; pos < NPOS; ++pos)
      for (int p = 0; p < 16; ++p)
        for (int co = 0; co < 64; ++co) {
          const double want = cur[((size_t)pos * 16 + p) * 64 + co], have = got[((size_t)p * 64 + co) * NPOS + pos];
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
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv_pixpair, dim3(nwg), dim3(512), lds_bytes, 0, dx, dw, (float *)nullptr, nwg, ds);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv_pixpair, dim3(nwg), dim3(512), lds_bytes, 0, dx, dw, (float *)nullptr, nwg, ds);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hs(nwg);
  hipMemcpy(hs.data(), ds, (size_t)nwg * 8, hipMemcpyDeviceToHost);
  std::sort(hs.begin(), hs.end());
  const double flop_useful = (double)rows * NCONV * 100.0 * 64 * 64 * 2, flop_padded = flop_useful * 1.44;
  printf("pixel-pair, %d rows (%d workgroups): %.4f ms per launch of %d convolutions; %.0f core cycles per workgroup pass (median); "
         "%.1f TFLOP/s of real products (x3 issued), %.1f TFLOP/s as the product kernel counts them (with padding)\n",
         rows, nwg, ms / reps, NCONV, (double)hs[nwg / 2], flop_useful * reps / (ms * 1e-3) / 1e12, flop_padded * reps / (ms * 1e-3) / 1e12);
  return 0;
}
