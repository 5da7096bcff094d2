// Experiment: do v_mfma_f32_32x32x16_f16 and v_cvt_f16_f32 keep fp16 SUBNORMAL inputs / results on gfx950?
// (decides whether a two-term fp16 operand split needs power-of-two pre-scaling of the residual terms)
//   hipcc --offload-arch=gfx950 -O2 tools/exp/f16_denorm.hip -o build_ab/f16_denorm && build_ab/f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float a_val, float b_val, float *out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)a_val; /* v_cvt_f16_f32 */
    b[i] = (_Float16)b_val;
  }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = acc[0];
    out[1] = (float)a[0];
    out[2] = (float)b[0];
  }
}

int main() {
  float *d;
  hipMalloc(&d, 64);
  const float cases[][2] = {{1.0f, 1.0f}, {0x1p-20f, 0x1p10f}, {0x1p-24f, 0x1p10f}, {0x1p-16f, 0x1p-16f}, {3e-6f, 1.0f}, {0x1p-14f, 1.0f}};
  for (auto &c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
    float h[3];
    hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("a = %.6e  b = %.6e : fp16(a) = %.6e fp16(b) = %.6e  mfma sum over k=16 = %.6e  (exact %.6e)\n", c[0], c[1], h[1], h[2], h[0],
           16.0 * (double)h[1] * (double)h[2]);
  }
  return 0;
}
