// nn_emu.cpp -- TEST INFRASTRUCTURE.  Plain CPU evaluation of mlp12x100 for the
// lane-loop emulation build (tests/emu), so that the engine's fused-mode HOST
// loop can be exercised without a GPU.  The product's network kernels are
// corintho_ai_amd/csrc/nn_mlp.hip / nn_rescnn.hip; nothing here ships.
#include <math.h>

#include <vector>

#include "../../corintho_ai_amd/csrc/engine_defs.h"
#include "../../corintho_ai_amd/csrc/nn.h"

struct EmuMlp : CoNet {
  std::vector<float> w;
  size_t cap;
  EmuMlp(const float *weights, size_t n, size_t max_rows) : w(weights, weights + n), cap(max_rows) {}
  size_t max_rows() const override { return cap; }
  int kind() const override { return CO_NET_MLP12X100; }
  double flop_per_row() const override { return 2.0 * (70 * 100 + 11 * 100 * 100 + 100 + 100 * 96); }
  void forward(const float *in, int32_t rows_cap, const int32_t *d_rows, float *ev, float *pr, rt_stream_t,
               const CoNetIO &io = CoNetIO()) override {
    int n = *d_rows;
    (void)rows_cap;
#pragma omp parallel for
    for (int r = 0; r < n; ++r) {
      float x[100], y[100];
      const float *p = w.data();
      int in_dim = 70;
      const size_t irow = io.in_idx ? (size_t)io.in_idx[r] : (size_t)r, orow = io.out_idx ? (size_t)io.out_idx[r] : (size_t)r;
      for (int i = 0; i < 70; ++i) x[i] = in[irow * CO_STATE_STRIDE + i];
      for (int l = 0; l < 12; ++l) {
        const float *K = p, *b = K + in_dim * 100, *ga = b + 100, *be = ga + 100, *mu = be + 100, *va = mu + 100;
        for (int o = 0; o < 100; ++o) {
          float s = 0.0f;
          for (int i = 0; i < in_dim; ++i) s = fmaf(x[i], K[i * 100 + o], s);
          s += b[o];
          s = s > 0.0f ? s : 0.0f;
          float a = (float)((double)ga[o] / sqrt((double)va[o] + CO_BN_EPS));
          float c = (float)((double)be[o] - (double)mu[o] * (double)a);
          y[o] = a * s + c;
        }
        for (int o = 0; o < 100; ++o) x[o] = y[o];
        p = va + 100;
        in_dim = 100;
      }
      const float *Kv = p, *bv = Kv + 100, *Kp = bv + 1, *bp = Kp + 9600;
      float v = 0.0f;
      for (int i = 0; i < 100; ++i) v = fmaf(x[i], Kv[i], v);
      ev[orow * (size_t)io.eval_stride] = tanhf(v + bv[0]);
      float lg[96], mx = -INFINITY;
      for (int o = 0; o < 96; ++o) {
        float s = 0.0f;
        for (int i = 0; i < 100; ++i) s = fmaf(x[i], Kp[i * 96 + o], s);
        lg[o] = s + bp[o];
        mx = lg[o] > mx ? lg[o] : mx;
      }
      float sum = 0.0f;
      for (int o = 0; o < 96; ++o) {
        lg[o] = expf(lg[o] - mx);
        sum += lg[o];
      }
      for (int o = 0; o < 96; ++o) pr[orow * (size_t)io.probs_stride + o] = lg[o] / sum;
    }
  }
};

CoNet *co_net_create(int kind, const float *weights, size_t n, size_t max_rows, rt_stream_t) {
  if (kind == CO_NET_MLP12X100 && n == (size_t)CO_MLP_NUM_WEIGHTS) return new EmuMlp(weights, n, max_rows);
  return nullptr;
}
