// nn.h -- policy/value network interface of the fused mode.
//
// A net evaluates the compact, game-major batch of leaf states produced by K4
// (rows of CO_STATE_STRIDE floats, 70 used) and writes one value and 96 prior
// probabilities per row, in the order the search reads them back
// (main.pyx:70-83 get_predictions: evals[:n] = res[0].flatten(); probs[:n] = res[1]).
// The row count lives on the device (req_offset[G]); kernels are launched for
// `rows_cap` rows and workgroups beyond the count exit, so the play loop never
// waits for the host.  Results of a row depend on that row only (fixed
// reduction order, no batch-dependent tiling): SURVEY 8e invariant.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "rt.h"

#define CO_NET_NUM_MOVES 96
#define CO_NET_MLP12X100 1
#define CO_NET_RESCNN4 2
#define CO_NET_RESCNN4_X3 3 /* same network and weights, convolutions at bf16x3 split precision */
#define CO_NET_MLP12X100_X3 4 /* mlp12x100, same weights, dense layers at bf16x3 split precision */
/* float32-equivalent arithmetic on the bf16 matrix pipe: both operands as three bf16 terms (their sum is the
 * float32 value), six MFMA products -- everything down to 2^-24 of a product is kept (nn_rescnn.hip) */
#define CO_NET_RESCNN4_X6 5
#define CO_NET_MLP12X100_X6 6
/* two fp16 terms per operand (22 significand bits), three MFMA products: float32-class arithmetic at the cost of bf16x3 */
#define CO_NET_RESCNN4_H3 8
#define CO_NET_MLP12X100_H3 9
/* (7 was round 2's Winograd experiment, git 22960a5:tools/exp/archive, removed from the tree in round 6) */

/* mlp12x100 flat weight layout (float32), matching Keras get_weights() order of
 * wrapper.py:256-271:
 *   for l in 0..11: kernel[in_l][100], bias[100], gamma[100], beta[100], moving_mean[100], moving_var[100]
 *   value head: kernel[100][1], bias[1];  policy head: kernel[100][96], bias[96]
 * in_0 = 70, in_l = 100.  BatchNormalization epsilon = 1e-3 (Keras default). */
#define CO_MLP_LAYERS 12
#define CO_MLP_WIDTH 100
#define CO_MLP_NUM_WEIGHTS (70 * 100 + 500 + 11 * (100 * 100 + 500) + 100 + 1 + 100 * 96 + 96)
#define CO_BN_EPS 1e-3

/* Optional indirection of a network launch (the evaluation cache of fused training): row r of the launch is read
 * from input row in_idx[r] and its outputs go to element out_idx[r] of arrays with the given strides (floats per
 * element).  All null / {1, 96}: rows in place, the reference layout. */
struct CoNetIO {
  const int32_t *in_idx = nullptr;
  const int32_t *out_idx = nullptr;
  int32_t eval_stride = 1;
  int32_t probs_stride = CO_NET_NUM_MOVES;
  /* Does this launch have the GPU to itself?  False when other streams keep it busy too (fused training in several pools:
   * the other pools' search and network kernels): a kernel family may then choose for throughput per CU rather than for
   * the latency of this launch (nn_rescnn.hip rcp_small_begin). */
  int32_t alone = 1;
};

struct CoNet {
  virtual ~CoNet() {}
  virtual size_t max_rows() const = 0;
  virtual int kind() const = 0;
  /* d_rows: device int32 holding the number of valid rows (<= rows_cap) */
  virtual void forward(const float *d_in, int32_t rows_cap, const int32_t *d_rows, float *d_eval, float *d_probs,
                       rt_stream_t s, const CoNetIO &io = CoNetIO()) = 0;
  /* algorithmic flop per row, for the roofline */
  virtual double flop_per_row() const = 0;
  /* The f16x3 kinds hold every operand as two fp16 terms: a folded weight or an activation beyond fp16's largest
   * finite value (CO_F16_MAX) would convert to infinity and the evaluation to NaN without a word.  Weights are checked
   * when the net is created (std::invalid_argument); the kernels track the largest activation they split and raise a
   * flag on the device, which this call reads (a 4-byte copy and a wait on `s`): true = some evaluation since the
   * net was created left the range, its outputs are not to be trusted -- the engine turns that into an error and names
   * the float32-equivalent x6 kind.  Kinds with float32's exponent range never report. */
  virtual bool range_exceeded(rt_stream_t) { return false; }
};
#define CO_F16_MAX 65504.0f

CoNet *co_net_create(int kind, const float *weights, size_t n_floats, size_t max_rows, rt_stream_t s);
