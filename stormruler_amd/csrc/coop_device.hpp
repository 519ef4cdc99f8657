// Device helpers shared by the persistent, grid-synchronising kernels (latency.hip: one kernel per solve for small
// operators, the Gram-Schmidt chain; resident.hip: block-resident lattice solves): coherent accesses, the tagged
// all-reduce slots, the wave sum.
#pragma once

#include "common.hpp"

namespace storm {

// Data that crosses wavefronts inside the kernel -- the published rows of r and p, the all-reduce slots -- is
// written and read with RELAXED AGENT-SCOPE ATOMIC accesses: single stores / loads that are coherent across the
// XCDs' private L2s (write-through, miss-through).  Whole-cache release / acquire fences (L2 write-back and
// invalidate, which an agent-scope fence means on this chip) are never issued: they cost ~100 us per barrier when
// 4 096 wavefronts execute them, and would evict the operator records, which are read-only and may stay cached.
__device__ __forceinline__ void co_store(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double co_load(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Publish one row for the other blocks' gathers.  xchg (the default): an atomic EXCHANGE whose returned value the wave
// consumes (`seen`, at the all-reduce that follows) -- a returning read-modify-write has been performed at the point
// of coherence, so the row is visible to every XCD before this block's all-reduce words go out, whatever else loads
// the memory system (csrc/ticket_device.hpp: an ACKNOWLEDGED write-through store was seen not yet visible to another
// XCD under a 256^3 streaming load).  xchg == 0: the write-through store, ordered by its acknowledgement only.
__device__ __forceinline__ void co_publish(double *p, double v, int xchg, unsigned long long &seen) {
  if (xchg)
    seen ^= __hip_atomic_exchange(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    co_store(p, v);
}

// All-reduce across a co-resident (cooperative) grid, which is also its barrier.  Slot (b, parity) of block b: 16
// bytes at slots + (2 b + (seq & 1)) * kLatSlotStride (256 bytes apart, so the polling load of all blocks spreads
// over the memory channels), holding the block's value in two self-validating 8-byte words
//     { low half of the double, tag }   { high half, tag }        (tag = low 32 bits of the sequence number)
// Each word is ONE 8-byte store -- atomic -- so the writer needs no ordering between them and no acknowledgement:
// two stores, fire and forget; a reader's 16-byte load is good when BOTH tags are the current one.  (The scheme of
// collective libraries' low-latency protocols.)  A synchronisation point costs: stores in flight, one polled load.
// TWO slots per block, used alternately: a block that has passed all-reduce `seq` may publish `seq + 1` while a
// slower block is still polling for `seq` -- into the other slot; it can only overwrite slot (seq & 1) with `seq + 2`
// after passing `seq + 1`, which needed the slow block's `seq + 1` words, which that block stores after it has
// finished reading `seq`.
constexpr int kLatSlotStride = 256;
constexpr long long kLatTimeoutTicks = 1000000000LL;  // 10 s of the 100 MHz real-time counter
__device__ __forceinline__ bool co_load_slot(const char *slot, unsigned tag, double *value) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(slot) : "memory");
  *value = __hiloint2double((int)w.z, (int)w.x);
  return w.y == tag && w.w == tag;
}
__device__ __forceinline__ void co_store_slot(char *slot, unsigned tag, double value) {
  const unsigned long long lo = ((unsigned long long)tag << 32) | (unsigned)__double2loint(value);
  const unsigned long long hi = ((unsigned long long)tag << 32) | (unsigned)__double2hiint(value);
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Two sums in one slot: four words.
__device__ __forceinline__ bool co_load_slot2(const char *slot, unsigned tag, double *v0, double *v1) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w0, w1;
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(w0), "=&v"(w1)
               : "v"(slot)
               : "memory");
  *v0 = __hiloint2double((int)w0.z, (int)w0.x);
  *v1 = __hiloint2double((int)w1.z, (int)w1.x);
  return w0.y == tag && w0.w == tag && w1.y == tag && w1.w == tag;
}
__device__ __forceinline__ double lat_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

}  // namespace storm
