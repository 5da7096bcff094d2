// nn_rescnn.hip -- K6: the north-star 4-block residual CNN (placeholder until
// the convolution kernels land; co_rescnn_create reports "not available").
#include "nn.h"

CoNet *co_rescnn_create(const float *, size_t, size_t, rt_stream_t) { return nullptr; }
