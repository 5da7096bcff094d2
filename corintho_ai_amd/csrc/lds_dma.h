// lds_dma.h -- weight streaming into LDS by LDS-DMA (global_load_lds_dwordx4) with the waits under
// the kernel's own control.
//
// The network kernels stream their weights through rings of LDS slots: while one slot is being
// multiplied, the next ones are in flight.  Written with __builtin_amdgcn_global_load_lds the
// compiler tracks every transfer and puts `s_waitcnt vmcnt(0)` in front of the first LDS read it
// cannot prove disjoint from a pending transfer (any read at a run-time offset), and
// __syncthreads() drains vmcnt as part of its fence -- both turn a prefetch into a stall.
// Here the transfer is an asm statement: the compiler sees neither a load nor an LDS store, and
// ordering is entirely explicit:
//     co_lds_dma_1k(...) ...                   request (counts in vmcnt, in issue order)
//     CO_WAIT_VMCNT(n)                         all but this wave's n youngest requests have landed
//     co_wg_barrier()                          ... and every other wave's too
//     LDS reads
// A slot may be requested again once every wave has passed a barrier behind its last read of it
// (a wave that reaches the barrier has issued the MFMAs that consumed those reads).
// M0 is written by these statements.  It cannot be named in their clobber lists (clang: "inline asm clobber
// list contains reserved registers: m0 ... may lead to undefined behaviour" -- M0 is a reserved register that
// the compiler never keeps live: it re-materialises M0 with an s_mov glued to each of its own instructions
// that read it, so a write in between is not observed); the statements are volatile and clobber memory, which
// keeps them in program order relative to each other and to the kernel's LDS accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

/* LDS byte address of a __shared__ object as the wave-uniform value M0 wants */
__device__ __forceinline__ uint32_t co_lds_addr(const void *p) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);
}

/* one wave-instruction = 1 KiB: lane i supplies the global address of bytes [16 i, 16 i + 16) of the
 * piece; they land at lds_byte_addr + 16 i (lds_byte_addr wave-uniform, 16-byte aligned) */
__device__ __forceinline__ void co_lds_dma_1k(const void *g_lane_ptr, uint32_t lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_addr), "v"(g_lane_ptr) : "memory");
}

#define CO_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

/* s_barrier without the fence of __syncthreads() (which would drain vmcnt): LDS data written by
 * DMA is ordered by the CO_WAIT_VMCNT in front of it */
__device__ __forceinline__ void co_wg_barrier() {
  /* this wave's LDS reads and writes have completed before it signals (what __syncthreads adds in front
   * of s_barrier, without its vmcnt drain): slot reuse does not rest on in-order LDS issue */
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
