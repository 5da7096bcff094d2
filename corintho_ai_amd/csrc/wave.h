// wave.h -- the wavefront programming layer of the MCTS kernels.
//
// The self-play kernels give ONE 64-lane wavefront to one game.  Control flow is
// wave-uniform by construction; lane parallelism appears only in short
// "FOR_LANES" sections (edge scans, legal-move generation, prior filtering,
// state expansion) separated by explicit cross-lane operations (ballot, max,
// broadcast).  This header names those few operations.
//
// Two builds of the same kernel source exist:
//   * the product: hipcc --offload-arch=gfx950.  LV(T,x) is a register, L(x) is
//     x, FOR_LANES runs its body once with `lane` = the hardware lane id, the
//     cross-lane macros are CDNA4 wave intrinsics (64-wide, DPP / ds_bpermute /
//     v_readlane).
//   * tests/emu (g++ -DCO_EMU): LV(T,x) is T x[64], FOR_LANES is a loop over
//     lanes.  It exists so that the kernel LOGIC can be tested and run under
//     the CPU sanitizers on a machine without a GPU.  It is test infrastructure and is
//     never loaded by the product.
//
// Rules for code written against this layer:
//   - a value used outside FOR_LANES is wave-uniform (identical in all lanes);
//   - inside one FOR_LANES section a lane never reads memory another lane
//     writes in the same section (no lockstep assumptions);
//   - cross-lane data moves only through the WAVE_* macros or through memory
//     between two sections.
#pragma once
#include <stdint.h>

#define CO_WAVE 64

/* Simulations of a step selected together by the search kernel (mcts.h co_search_rows): 4 = one per row of the
 * wavefront, 1 = one after another (rounds 1-4). */
#ifndef CO_SB
#define CO_SB 4
#endif

#ifdef CO_EMU
// ------------------------------------------------------------------ emulation
#include <math.h>
#include <string.h>
struct uint4 {
  uint32_t x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  uint4 v = {x, y, z, w};
  return v;
}
#define CO_DEV static inline
#define CO_COLD static /* the product build keeps these out of line (see below) */
#define CO_COLD2 static
#define CO_KERNEL static void
#define CO_CONST static const
#define LV(T, x) T x[CO_WAVE]
#define LVP(T, x) T *x /* an LV variable as a function parameter */
#define LVPA(T, x, N) T (*x)[CO_WAVE] /* ... an array of N of them */
#define CO_OPAQUE_V(x) ((void)0)
#define L(x) x[lane]
#define LAT(x, i) x[(i)]
#define FOR_LANES for (int lane = 0; lane < CO_WAVE; ++lane)
#define FOR_LANES_HOT FOR_LANES
#define WAVE_SHARED(T, x, n) T x[n]
#define WG_SHARED(T, x, n) T x[n] /* one copy for the workgroup's wavefronts (here: the one wavefront) */
#define WAVE_SYNC() ((void)0)
#define UNI_I(x) (x)
#define UNI_U(x) (x)
#define UNI_F(x) (x)

static inline uint64_t emu_ballot(const int *p) {
  uint64_t m = 0;
  for (int i = 0; i < CO_WAVE; ++i)
    if (p[i]) m |= 1ull << i;
  return m;
}
static inline float emu_max_f32(const float *p) {
  float m = p[0];
  for (int i = 1; i < CO_WAVE; ++i) m = p[i] > m ? p[i] : m;
  return m;
}
static inline int emu_sum_i32(const int *p) {
  int s = 0;
  for (int i = 0; i < CO_WAVE; ++i) s += p[i];
  return s;
}
#define WAVE_BALLOT(p) emu_ballot(p)
#define WAVE_MAX_F32(x) emu_max_f32(x)
#define WAVE_SUM_I32(x) emu_sum_i32(x)
#define WAVE_BCAST(x, i) (x[(i)])
/* inside FOR_LANES: the number of set bits of a wave-uniform 64-bit mask below this lane's bit */
#define LANE_RANK64(m) __builtin_popcountll((unsigned long long)(m) & ((1ull << lane) - 1ull))
static inline int co_popc64(uint64_t v) { return __builtin_popcountll(v); }
static inline int co_popc32(uint32_t v) { return __builtin_popcount(v); }
static inline int co_ffs64(uint64_t v) { return __builtin_ffsll((long long)v); } /* 1-based, 0 if none */
static inline double co_sqrt_f64(double x) { return sqrt(x); }
static inline unsigned long long co_atomic_add_u64(unsigned long long *p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
static inline void co_atomic_add_u64_noret(unsigned long long *p, unsigned long long v) { (void)__atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
/* 32-bit atomics of the evaluation cache (kernels.h): compare-and-swap returning the old value, add, and a load /
 * store that other workgroups' atomics are coherent with */
static inline uint32_t co_atomic_cas_u32(uint32_t *p, uint32_t expected, uint32_t desired) {
  __atomic_compare_exchange_n(p, &expected, desired, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
  return expected;
}
static inline uint32_t co_atomic_add_u32(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t co_atomic_load_u32(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline void co_atomic_store_u32(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static inline void co_wait_stores() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
/* per-lane coherent accesses (inside FOR_LANES) */
static inline uint32_t co_lane_load_coherent_u32(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline void co_lane_store_coherent_u32(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static inline uint32_t co_lane_cas_u32(uint32_t *p, uint32_t expected, uint32_t desired) {
  __atomic_compare_exchange_n(p, &expected, desired, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
  return expected;
}
/* ---- rows: a wavefront as four groups of 16 lanes (the rows of the DPP unit).  Inside a row every lane may hold
 * the same value ("row-uniform"); the ROW_* statements below move data inside each row, all four rows at once.
 * They are statements used OUTSIDE FOR_LANES (dst and src are LV variables). */
#define CO_ROW_LANES 16
static inline void emu_row_max_f32(float *d, const float *s) {
  float t[CO_WAVE];
  for (int r = 0; r < CO_WAVE; r += CO_ROW_LANES) {
    float m = s[r];
    for (int i = 1; i < CO_ROW_LANES; ++i) m = s[r + i] > m ? s[r + i] : m;
    for (int i = 0; i < CO_ROW_LANES; ++i) t[r + i] = m;
  }
  memcpy(d, t, sizeof t);
}
static inline void emu_row_min_u32(uint32_t *d, const uint32_t *s) {
  uint32_t t[CO_WAVE];
  for (int r = 0; r < CO_WAVE; r += CO_ROW_LANES) {
    uint32_t m = s[r];
    for (int i = 1; i < CO_ROW_LANES; ++i) m = s[r + i] < m ? s[r + i] : m;
    for (int i = 0; i < CO_ROW_LANES; ++i) t[r + i] = m;
  }
  memcpy(d, t, sizeof t);
}
static inline void emu_row_ballot(uint32_t *d, const int *p) {
  for (int r = 0; r < CO_WAVE; r += CO_ROW_LANES) {
    uint32_t m = 0;
    for (int i = 0; i < CO_ROW_LANES; ++i)
      if (p[r + i]) m |= 1u << i;
    for (int i = 0; i < CO_ROW_LANES; ++i) d[r + i] = m;
  }
}
static inline void emu_row_shfl_u32(uint32_t *d, const uint32_t *s, const int *col) {
  uint32_t t[CO_WAVE];
  for (int l = 0; l < CO_WAVE; ++l) t[l] = s[(l & ~(CO_ROW_LANES - 1)) + (col[l] & (CO_ROW_LANES - 1))];
  memcpy(d, t, sizeof t);
}
static inline void emu_wave_shfl_u32(uint32_t *d, const uint32_t *s, const int *src) {
  uint32_t t[CO_WAVE];
  for (int l = 0; l < CO_WAVE; ++l) t[l] = s[src[l] & (CO_WAVE - 1)];
  memcpy(d, t, sizeof t);
}
static inline void emu_row_sum_i32(int *d, const int *s) {
  int t[CO_WAVE];
  for (int r = 0; r < CO_WAVE; r += CO_ROW_LANES) {
    int m = 0;
    for (int i = 0; i < CO_ROW_LANES; ++i) m += s[r + i];
    for (int i = 0; i < CO_ROW_LANES; ++i) t[r + i] = m;
  }
  memcpy(d, t, sizeof t);
}
/* acc += v of column 0, then of column 1, ... column 15 of the lane's row: sixteen SEQUENTIAL float additions (the
 * reference's loops over edges add in edge order) */
static inline void emu_row_seq_sum16(float *acc, const float *v) {
  for (int r = 0; r < CO_WAVE; r += CO_ROW_LANES) {
    float a = acc[r];
    for (int i = 0; i < CO_ROW_LANES; ++i) a += v[r + i];
    for (int i = 0; i < CO_ROW_LANES; ++i) acc[r + i] = a;
  }
}
#define ROW_SUM_I32(d, s) emu_row_sum_i32(d, s)
#define ROW_SEQ_SUM16(acc, v) emu_row_seq_sum16(acc, v)
#define ROW_SEQ_SUM16_2(a0, v0, a1, v1) (emu_row_seq_sum16(a0, v0), emu_row_seq_sum16(a1, v1))
#define ROW_MAX_F32(d, s) emu_row_max_f32(d, s)
#define ROW_MIN_U32(d, s) emu_row_min_u32(d, s)
#define ROW_BALLOT(d, p) emu_row_ballot(d, p)           /* d = the 16 predicate bits of the lane's row */
#define ROW_SHFL_U32(d, s, col) emu_row_shfl_u32(d, s, col) /* d = s of lane `col` (an LV int, 0..15) of the same row */
#define WAVE_SHFL_U32(d, s, src) emu_wave_shfl_u32(d, s, src) /* d = s of lane `src` (an LV int, 0..63) */
extern thread_local int co_emu_block_idx;
#define CO_BLOCK_IDX co_emu_block_idx
/* kernels whose wavefronts are independent may pack several per workgroup on the GPU (fewer, fatter
 * workgroups for the dispatcher); the emulation runs one wavefront per block */
#define CO_WAVES_PER_BLOCK 1
#define CO_WAVE_IN_BLOCK 0

#else
// ------------------------------------------------------------------- gfx950
#include <hip/hip_runtime.h>
#define CO_DEV __device__ __forceinline__
/* once-per-ply and rare paths (move choice, the sequential simulation, slot recycling).  Round 5 measured them as real calls
 * (-DCO_COLD_NOINLINE: the search kernel's spills go from 328 scalar + 36 vector registers to 54 + 58, its scratch from
 * 240 to 1208 bytes per lane): a generation of 4096 games with the reference's network takes 262 ms instead of 181 --
 * every call saves and restores through scratch memory.  Inlined, as before. */
#ifdef CO_COLD_NOINLINE
#define CO_COLD __device__ __attribute__((noinline))
#define CO_COLD2 __device__ __attribute__((noinline))
#else
#define CO_COLD __device__ __forceinline__
#define CO_COLD2 __device__ __forceinline__
#endif
#define CO_KERNEL __global__ void
#define CO_CONST __device__ const
#define LV(T, x) T x
#define LVP(T, x) T &x
#define LVPA(T, x, N) T (&x)[N]
/* the compiler may not look through x from here on.  For values derived from the lane index in hot sections: left
 * transparent, every such expression (1 << column, row * stride, ...) is hoisted to the kernel's entry, lives across the
 * whole step and is spilled -- and a reload from scratch behind a store waits for the store (round 5: six of them in the
 * node-store loop of the grouped search cost more than the loop) */
#define CO_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define L(x) x
#define LAT(x, i) x
/* the same with a lane index the compiler cannot see through (CO_OPAQUE_V): nothing derived from it leaves the section */
__device__ __forceinline__ int co_lane_opaque() {
  int l = (int)(threadIdx.x & 63);
  CO_OPAQUE_V(l);
  return l;
}
#define FOR_LANES_HOT for (int lane = co_lane_opaque(), _co_once = 1; _co_once; _co_once = 0)
#ifdef CO_LANES_TRANSPARENT
#define FOR_LANES for (int lane = (int)(threadIdx.x & 63), _co_once = 1; _co_once; _co_once = 0)
#else
#define FOR_LANES FOR_LANES_HOT
#endif
/* Wavefronts of the search kernel per workgroup.  A game is one wavefront whatever this is; what it changes is where
 * the dispatcher puts them: single-wave workgroups spread a launch of 2048 games over all 1024 SIMDs at two waves
 * each, CO_K3_WPB = 16 packs them onto half of the CUs at four waves per SIMD and leaves the other CUs to the
 * network kernel of the other pool (whose waves take a SIMD's whole register file). */
#ifndef CO_K3_WPB
#define CO_K3_WPB 16 /* measured (round 3, rescnn4h3, two pools): 1 -> 466.7 ms, 4 -> 463.6, 16 -> 455.6 per generation; the MLP: +-0.
                     * Round 5, three pools, grouped search: 16 / 8 / 4 -> 397.2 / 397.2 / 398.8 (rescnn4h3), 130.3 / 130.2 / 129.8 (mlp12x100h3): the
                     * search's device time falls 5 % with fewer wavefronts per SIMD and the network's rises by as much */
#endif
#if CO_K3_WPB == 1
#define WAVE_SHARED(T, name_, n) __shared__ T name_[n]
#else
#define WAVE_SHARED(T, name_, n)              \
  __shared__ T name_##_blk[CO_K3_WPB][n];     \
  T(&name_)[n] = name_##_blk[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))]
#endif
#define WG_SHARED(T, name_, n) __shared__ T name_[n]
// a wave barrier orders LDS traffic of the wave (its LDS slices are its own)
#define WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#define UNI_I(x) __builtin_amdgcn_readfirstlane((int)(x))
#define UNI_U(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define UNI_F(x) __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)))

/* Cross-lane reductions stay in the VALU: four DPP steps reduce each row of 16
 * lanes (quad_perm xor-1, xor-2, row_half_mirror, row_mirror -- valid for
 * commutative ops because every lane of a group already holds the group's
 * value), then four v_readlane + scalar-operand ops join the rows.  No LDS
 * round trips (ds_bpermute), which dominated the search loop's latency. */
#define CO_DPP_XOR1 0xB1        /* quad_perm [1,0,3,2] */
#define CO_DPP_XOR2 0x4E        /* quad_perm [2,3,0,1] */
#define CO_DPP_HALF_MIRROR 0x141
#define CO_DPP_MIRROR 0x140
#define CO_DPP_I(v, ctrl) __builtin_amdgcn_mov_dpp((v), (ctrl), 0xF, 0xF, false)
#define CO_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (ctrl), 0xF, 0xF, false))
/* Wave maximum of a float that is never NaN (PUCT values: finite or -inf), result wave-uniform and bit-equal to one
 * lane's input (+0 / -0 aside, which compare equal).  Six v_max_f32 with a DPP source: four steps inside each row of 16
 * lanes, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 leave the wave's maximum in lane 63.
 * (As v_mov_dpp + v_cmp + v_cndmask per step, then four v_readlane and scalar selects, the reduction was ~30 dependent
 * instructions in the middle of every PUCT scan.)  The s_nop 1 in front of each step are the two wait states a DPP read
 * needs behind the VALU write of its source, which the assembler does not insert. */
__device__ __forceinline__ float co_wave_max_f32(float v) {
  float r;
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "=&v"(r)
      : "v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 63));
}
__device__ __forceinline__ int co_wave_sum_i32(int v) {
  v += CO_DPP_I(v, CO_DPP_XOR1);
  v += CO_DPP_I(v, CO_DPP_XOR2);
  v += CO_DPP_I(v, CO_DPP_HALF_MIRROR);
  v += CO_DPP_I(v, CO_DPP_MIRROR);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}
/* broadcast from a wave-uniform lane index: v_readlane, not ds_bpermute */
__device__ __forceinline__ int co_rl(int v, int i) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(i)); }
template <typename T>
__device__ __forceinline__ T co_bcast(T v, int i);
template <>
__device__ __forceinline__ int co_bcast<int>(int v, int i) {
  return co_rl(v, i);
}
template <>
__device__ __forceinline__ uint32_t co_bcast<uint32_t>(uint32_t v, int i) {
  return (uint32_t)co_rl((int)v, i);
}
template <>
__device__ __forceinline__ float co_bcast<float>(float v, int i) {
  return __int_as_float(co_rl(__float_as_int(v), i));
}
template <>
__device__ __forceinline__ uint4 co_bcast<uint4>(uint4 v, int i) {
  uint4 r;
  int u = __builtin_amdgcn_readfirstlane(i);
  r.x = (uint32_t)__builtin_amdgcn_readlane((int)v.x, u);
  r.y = (uint32_t)__builtin_amdgcn_readlane((int)v.y, u);
  r.z = (uint32_t)__builtin_amdgcn_readlane((int)v.z, u);
  r.w = (uint32_t)__builtin_amdgcn_readlane((int)v.w, u);
  return r;
}
#define WAVE_BALLOT(p) ((uint64_t)__ballot(p))
#define WAVE_MAX_F32(x) co_wave_max_f32(x)
#define WAVE_SUM_I32(x) co_wave_sum_i32(x)
#define WAVE_BCAST(x, i) co_bcast(x, (i))
/* inside FOR_LANES: the number of set bits of a wave-uniform 64-bit mask below this lane's bit (v_mbcnt_lo / _hi) */
#define LANE_RANK64(m) ((int)__builtin_amdgcn_mbcnt_hi((uint32_t)((uint64_t)(m) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u)))
__device__ __forceinline__ int co_popc64(uint64_t v) { return __popcll(v); }
__device__ __forceinline__ int co_popc32(uint32_t v) { return __popc(v); }
__device__ __forceinline__ int co_ffs64(uint64_t v) { return __ffsll((unsigned long long)v); }
__device__ __forceinline__ double co_sqrt_f64(double x) { return __builtin_sqrt(x); }
/* one device-scope atomic per wave (lane 0), old value broadcast to the wave */
__device__ __forceinline__ unsigned long long co_atomic_add_u64(unsigned long long *p, unsigned long long v) {
  unsigned long long old = 0;
  if ((threadIdx.x & 63) == 0) old = atomicAdd(p, v);
  unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)old);
  unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(old >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
/* the same, nobody waits for the old value */
__device__ __forceinline__ void co_atomic_add_u64_noret(unsigned long long *p, unsigned long long v) {
  if ((threadIdx.x & 63) == 0) (void)atomicAdd(p, v);
}
/* 32-bit device-scope atomics of the evaluation cache (kernels.h), one per wave (lane 0), result broadcast.  The
 * 8 XCDs of the chip have separate L2s: a header word another workgroup may be writing in the same launch is
 * read and written through these (device-scope, performed at the memory side), never through a plain access. */
__device__ __forceinline__ uint32_t co_atomic_cas_u32(uint32_t *p, uint32_t expected, uint32_t desired) {
  uint32_t old = 0;
  if ((threadIdx.x & 63) == 0) old = atomicCAS(p, expected, desired);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
}
__device__ __forceinline__ uint32_t co_atomic_add_u32(uint32_t *p, uint32_t v) {
  uint32_t old = 0;
  if ((threadIdx.x & 63) == 0) old = atomicAdd(p, v);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
}
/* RELAXED on purpose: an acquire / release at device scope makes the wave invalidate / write back its XCD's whole L2
 * (buffer_inv / buffer_wbl2 sc1) -- measured: the search kernels three times slower with thousands of such waves in
 * flight.  The accesses themselves are performed at the coherent level (sc1); the cache's protocol tolerates any order
 * in which its words become visible (kernels.h). */
__device__ __forceinline__ uint32_t co_atomic_load_u32(const uint32_t *p) {
  uint32_t v = 0;
  if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void co_atomic_store_u32(uint32_t *p, uint32_t v) {
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* this wave's earlier stores have been acknowledged (no cache maintenance) */
__device__ __forceinline__ void co_wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
/* per-lane device-scope accesses (inside FOR_LANES): every lane its own word, relaxed (see above) */
__device__ __forceinline__ uint32_t co_lane_load_coherent_u32(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void co_lane_store_coherent_u32(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t co_lane_cas_u32(uint32_t *p, uint32_t expected, uint32_t desired) { return atomicCAS(p, expected, desired); }
/* ---- rows: a wavefront as four groups of 16 lanes (the rows of the DPP unit).  Inside a row every lane may hold
 * the same value ("row-uniform"); the ROW_* statements move data inside each row, all four rows at once. */
#define CO_ROW_LANES 16
__device__ __forceinline__ float co_row_max_f32(float v) {
  float r;
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
      : "=&v"(r)
      : "v"(v));
  return r;
}
__device__ __forceinline__ uint32_t co_row_min_u32(uint32_t v) {
  uint32_t r;
  asm("s_nop 1\n\t"
      "v_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
      : "=&v"(r)
      : "v"(v));
  return r;
}
__device__ __forceinline__ uint32_t co_row_ballot(int p) {
  const unsigned long long m = __ballot(p);
  return (uint32_t)(m >> (threadIdx.x & 48u)) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t co_row_shfl_u32(uint32_t v, int col) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 48u) | ((unsigned)col & 15u)) << 2), (int)v);
}
__device__ __forceinline__ int co_row_sum_i32(int v) {
  v += CO_DPP_I(v, CO_DPP_XOR1);
  v += CO_DPP_I(v, CO_DPP_XOR2);
  v += CO_DPP_I(v, CO_DPP_HALF_MIRROR);
  v += CO_DPP_I(v, CO_DPP_MIRROR);
  return v;
}
/* sixteen v_add_f32 with a DPP source (row_newbcast:c = column c of the lane's own row), one after the other.  Written as
 * the instructions themselves: from the builtin (update_dpp, then an addition) the compiler makes a v_mov_b32 of the "old"
 * value, a v_mov_b32_dpp and a packed addition per column -- five instructions where two chains need two, and a step's
 * priors are issue-bound there (round 5: 3.0 k of a pass's 9 k cycles).  The leading s_nop covers the two wait states
 * between a VALU write of the DPP source and its DPP read, which the hazard recogniser cannot see into the asm for. */
/* (one instruction per column, generated: the sixteen columns of a row in order) */
#define CO_NEWBCAST_ADD(D, S, C) "v_add_f32_dpp " D ", " S ", " D " row_newbcast:" #C " row_mask:0xf bank_mask:0xf\n\t"
#define CO_FOR_16_COLUMNS(X)                                                                                              \
  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
__device__ __forceinline__ float co_row_seq_sum16(float a, float v) {
#define CO_SUM1(C) CO_NEWBCAST_ADD("%0", "%1", C)
  asm("s_nop 1\n\t" CO_FOR_16_COLUMNS(CO_SUM1) : "+v"(a) : "v"(v));
#undef CO_SUM1
  return a;
}
/* two independent sums, interleaved column by column */
__device__ __forceinline__ void co_row_seq_sum16_2(float &a0, float v0, float &a1, float v1) {
#define CO_SUM2(C) CO_NEWBCAST_ADD("%0", "%2", C) CO_NEWBCAST_ADD("%1", "%3", C)
  asm("s_nop 1\n\t" CO_FOR_16_COLUMNS(CO_SUM2) : "+v"(a0), "+v"(a1) : "v"(v0), "v"(v1));
#undef CO_SUM2
}
#define ROW_SUM_I32(d, s) ((d) = co_row_sum_i32(s))
#define ROW_SEQ_SUM16(acc, v) ((acc) = co_row_seq_sum16((acc), (v)))
#define ROW_SEQ_SUM16_2(a0, v0, a1, v1) co_row_seq_sum16_2((a0), (v0), (a1), (v1))
#define ROW_MAX_F32(d, s) ((d) = co_row_max_f32(s))
#define ROW_MIN_U32(d, s) ((d) = co_row_min_u32(s))
#define ROW_BALLOT(d, p) ((d) = co_row_ballot(p))
#define ROW_SHFL_U32(d, s, col) ((d) = co_row_shfl_u32((s), (col)))
#define WAVE_SHFL_U32(d, s, src) ((d) = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((unsigned)(src) & 63u) << 2), (int)(s)))
#define CO_BLOCK_IDX ((int)blockIdx.x)
#define CO_WAVES_PER_BLOCK 4
#define CO_WAVE_IN_BLOCK ((int)(threadIdx.x >> 6))
#endif

CO_DEV float co_u2f(uint32_t u) {
#ifdef CO_EMU
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __uint_as_float(u);
#endif
}
CO_DEV uint32_t co_f2u(float f) {
#ifdef CO_EMU
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
#else
  return __float_as_uint(f);
#endif
}
