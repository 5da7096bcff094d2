// rt.h -- the few runtime calls the host engine needs.
// Product build: HIP runtime (streams, events, async copies).
// tests/emu build (-DCO_EMU): plain host memory and a block loop, so the same
// engine code runs the same kernel source on a machine without a GPU.
#pragma once
#include <stdexcept>
#include <string>

#include "wave.h"

#ifdef CO_EMU
#include <stdlib.h>
#include <string.h>
typedef int rt_stream_t;
struct rt_event_t {
  int dummy;
};
inline void rt_set_device(int) {}
inline void rt_malloc(void **p, size_t n, rt_stream_t) {
  *p = calloc(1, n ? n : 1);
  if (!*p) throw std::runtime_error("emu: out of memory");
}
inline void rt_free(void *p) { free(p); }
inline size_t rt_mem_free() { return (size_t)1 << 46; } /* host memory: no device budget to respect */
inline void rt_h2d(void *d, const void *h, size_t n, rt_stream_t) { memcpy(d, h, n); }
inline void rt_d2h(void *h, const void *d, size_t n, rt_stream_t) { memcpy(h, d, n); }
inline void rt_d2d(void *d, const void *s, size_t n, rt_stream_t) { memcpy(d, s, n); }
inline void rt_d2h_2d(void *h, size_t hpitch, const void *d, size_t dpitch, size_t width, size_t rows, rt_stream_t) {
  for (size_t r = 0; r < rows; ++r) memcpy((char *)h + r * hpitch, (const char *)d + r * dpitch, width);
}
inline void rt_memset(void *d, int v, size_t n, rt_stream_t) { memset(d, v, n); }
inline void rt_sync(rt_stream_t) {}
inline void rt_stream_create(rt_stream_t *s) { *s = 0; }
inline void rt_stream_destroy(rt_stream_t) {}
inline void rt_event_create(rt_event_t *) {}
inline void rt_event_destroy(rt_event_t) {}
inline void rt_event_record(rt_event_t, rt_stream_t) {}
inline void rt_event_sync(rt_event_t) {}
inline void rt_stream_wait(rt_stream_t, rt_event_t) {}
inline float rt_event_elapsed_ms(rt_event_t, rt_event_t) { return 0.0f; }
inline void rt_host_alloc(void **p, size_t n) { rt_malloc(p, n, 0); }
inline void rt_host_free(void *p) { free(p); }
inline bool rt_host_register(void *, size_t) { return true; }
inline void rt_host_unregister(void *) {}
#define RT_LAUNCH(kernel, grid, block, stream, ...)      \
  do {                                                   \
    int _grid = (int)(grid);                             \
    _Pragma("omp parallel for schedule(dynamic, 1)")     \
    for (int _b = 0; _b < _grid; ++_b) {                 \
      co_emu_block_idx = _b;                             \
      kernel(__VA_ARGS__);                               \
    }                                                    \
  } while (0)
#else
#include <hip/hip_runtime.h>
typedef hipStream_t rt_stream_t;
typedef hipEvent_t rt_event_t;
inline void rt_check(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
#define RT_CHECK(x) rt_check((x), #x)
inline void rt_set_device(int d) { RT_CHECK(hipSetDevice(d)); }
/* zeroed device memory; the clear is queued on the caller's stream `s` (the engine's streams are
 * non-blocking, so a clear on the null stream would not be ordered with them) and the caller's later
 * work on `s` -- or its next rt_sync(s) before other streams touch the buffer -- is ordered behind it */
inline void rt_malloc(void **p, size_t n, hipStream_t s) {
  RT_CHECK(hipMalloc(p, n ? n : 1));
  RT_CHECK(hipMemsetAsync(*p, 0, n ? n : 1, s));
}
inline void rt_free(void *p) {
  if (p) (void)hipFree(p);
}
inline size_t rt_mem_free() {
  size_t f = 0, t = 0;
  RT_CHECK(hipMemGetInfo(&f, &t));
  return f;
}
inline void rt_h2d(void *d, const void *h, size_t n, rt_stream_t s) {
  if (n) RT_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
}
inline void rt_d2h(void *h, const void *d, size_t n, rt_stream_t s) {
  if (n) RT_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
}
/* `rows` pieces of `width` bytes, `dpitch` apart on the device, to pieces `hpitch` apart on the host */
inline void rt_d2h_2d(void *h, size_t hpitch, const void *d, size_t dpitch, size_t width, size_t rows, rt_stream_t s) {
  if (rows && width) RT_CHECK(hipMemcpy2DAsync(h, hpitch, d, dpitch, width, rows, hipMemcpyDeviceToHost, s));
}
inline void rt_d2d(void *d, const void *s_, size_t n, rt_stream_t s) {
  if (n) RT_CHECK(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s));
}
inline void rt_memset(void *d, int v, size_t n, rt_stream_t s) {
  if (n) RT_CHECK(hipMemsetAsync(d, v, n, s));
}
inline void rt_sync(rt_stream_t s) { RT_CHECK(hipStreamSynchronize(s)); }
inline void rt_stream_create(rt_stream_t *s) { RT_CHECK(hipStreamCreateWithFlags(s, hipStreamNonBlocking)); }
#ifdef CO_EXP_CU_MASK
/* diagnostic builds only: stream `p` of `n` restricted to a share of the 256 compute units.  mode 1: a contiguous range of
 * mask bits, mode 2: every n-th bit, mode 3: all bits (the masked API with nothing masked: the control) */
inline void rt_stream_create_masked(rt_stream_t *s, int p, int n, int mode) {
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int cu = 0; cu < 256; ++cu) {
    const bool mine = mode == 1 ? (cu * n / 256 == p) : mode == 2 ? (cu % n == p) : true;
    if (mine) mask[cu >> 5] |= 1u << (cu & 31);
  }
  RT_CHECK(hipExtStreamCreateWithCUMask(s, 8, mask));
}
#endif
inline void rt_stream_destroy(rt_stream_t s) {
  if (s) (void)hipStreamDestroy(s);
}
inline void rt_event_create(rt_event_t *e) { RT_CHECK(hipEventCreate(e)); }
inline void rt_event_destroy(rt_event_t e) {
  if (e) (void)hipEventDestroy(e);
}
inline void rt_event_record(rt_event_t e, rt_stream_t s) { RT_CHECK(hipEventRecord(e, s)); }
inline void rt_event_sync(rt_event_t e) { RT_CHECK(hipEventSynchronize(e)); }
/* everything queued on s from here on starts behind what was queued in front of the record of e */
inline void rt_stream_wait(rt_stream_t s, rt_event_t e) { RT_CHECK(hipStreamWaitEvent(s, e, 0)); }
inline float rt_event_elapsed_ms(rt_event_t a, rt_event_t b) {
  float ms = 0.0f;
  RT_CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}
inline void rt_host_alloc(void **p, size_t n) { RT_CHECK(hipHostMalloc(p, n ? n : 1, hipHostMallocDefault)); }
inline void rt_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}
/* page-lock caller memory so that copies from / to it are direct DMA (no staging through the
 * runtime's bounce buffers); false = the runtime refused (the copy still works, pageable) */
inline bool rt_host_register(void *p, size_t n) {
  if (hipHostRegister(p, n, hipHostRegisterDefault) == hipSuccess) return true;
  (void)hipGetLastError();
  return false;
}
inline void rt_host_unregister(void *p) {
  if (p) (void)hipHostUnregister(p);
}
#define RT_LAUNCH(kernel, grid, block, stream, ...)                                           \
  do {                                                                                        \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, stream, __VA_ARGS__); \
    RT_CHECK(hipGetLastError());                                                              \
  } while (0)
#endif
