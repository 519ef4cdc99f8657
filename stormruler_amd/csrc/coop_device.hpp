// Device helpers shared by the persistent, grid-synchronising kernels (latency.hip: one kernel per solve for small
// operators, the Gram-Schmidt chain; resident.hip: block-resident lattice solves): coherent accesses, the tagged
// all-reduce slots, the wave sum.
#pragma once

#include "common.hpp"
#include "wave_device.hpp"

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
constexpr int kLatSlotStride = 64;
// Slot of block b for all-reduce `seq`: the blocks' slots of one parity are CONTIGUOUS (round 3: 512 bytes apart, "so
// that the polling spreads over the memory channels" -- but every block polls every slot, and 256 threads reading 256
// scattered lines are 256 requests per poll and block where 256 x 64 contiguous bytes are 128; under a streaming load
// the requests are what a synchronisation point waits for).
__device__ __forceinline__ size_t lat_slot_offset(unsigned block, unsigned long long seq) {
  return ((size_t)(seq & 1) * 256 + block) * kLatSlotStride;
}
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
// Three sums in one slot: six words, ONE round trip (round 3 polled them with two dependent loads -- the paired
// Gram-Schmidt chain's synchronisation point cost two memory round trips per poll instead of one).
__device__ __forceinline__ bool co_load_slot3(const char *slot, unsigned tag, double *v0, double *v1, double *v2) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w0, w1, w2;
  asm volatile(
      "global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(w0), "=&v"(w1), "=&v"(w2)
      : "v"(slot)
      : "memory");
  *v0 = __hiloint2double((int)w0.z, (int)w0.x);
  *v1 = __hiloint2double((int)w1.z, (int)w1.x);
  *v2 = __hiloint2double((int)w2.z, (int)w2.x);
  return w0.y == tag && w0.w == tag && w1.y == tag && w1.w == tag && w2.y == tag && w2.w == tag;
}
__device__ __forceinline__ double lat_wave_sum(double v) {
  return wave_sum_all(v);  // (= the xor butterfly's value in every lane, bit for bit: wave_device.hpp)
}


// N sums (N <= 10) over a co-resident grid of blocks of WAVES wavefronts, in slots `stride` bytes apart (parity-major:
// slot of block b at ((seq & 1) * 256 + b) * stride); every value travels as two self-validating words like the
// one- to three-value forms above, all of a slot's words are polled in ONE round trip.  Identical bits in every thread of
// every block (fixed folding order); bounded wait -> *gave_up.
template <int NV>
__device__ __forceinline__ bool co_load_slot_n(const char *slot, unsigned tag, double (&v)[NV]) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 w[NV];
  // (one base register, immediate offsets; nothing is consumed before the last load is in flight)
  if constexpr (NV == 10) {
    asm volatile(
        "global_load_dwordx4 %0, %10, off sc1\n\tglobal_load_dwordx4 %1, %10, off offset:16 sc1\n\t"
        "global_load_dwordx4 %2, %10, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %10, off offset:48 sc1\n\t"
        "global_load_dwordx4 %4, %10, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %10, off offset:80 sc1\n\t"
        "global_load_dwordx4 %6, %10, off offset:96 sc1\n\tglobal_load_dwordx4 %7, %10, off offset:112 sc1\n\t"
        "global_load_dwordx4 %8, %10, off offset:128 sc1\n\tglobal_load_dwordx4 %9, %10, off offset:144 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]), "=&v"(w[8]), "=&v"(w[9])
        : "v"(slot)
        : "memory");
  } else if constexpr (NV == 6) {
    asm volatile(
        "global_load_dwordx4 %0, %6, off sc1\n\tglobal_load_dwordx4 %1, %6, off offset:16 sc1\n\t"
        "global_load_dwordx4 %2, %6, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %6, off offset:48 sc1\n\t"
        "global_load_dwordx4 %4, %6, off offset:64 sc1\n\tglobal_load_dwordx4 %5, %6, off offset:80 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5])
        : "v"(slot)
        : "memory");
  } else if constexpr (NV == 3) {
    asm volatile(
        "global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
        "global_load_dwordx4 %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2])
        : "v"(slot)
        : "memory");
  } else if constexpr (NV == 2) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1])
                 : "v"(slot)
                 : "memory");
  } else {
    static_assert(NV == 1, "co_load_slot_n: 1, 2, 3, 6 or 10 values");
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w[0]) : "v"(slot) : "memory");
  }
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j] = __hiloint2double((int)w[j].z, (int)w[j].x);
    ok = ok && w[j].y == tag && w[j].w == tag;
  }
  return ok;
}
template <int NV, int WAVES>
__device__ __forceinline__ void co_allreduce_n(double (&s)[NV], char *slots, int stride, int *gave_up, unsigned long long seq,
                                               double *lds /* [NV * WAVES] */) {
  static_assert(NV == 1 || NV == 6 || NV == 10, "co_allreduce_n: the wait statement lists its registers");
  const unsigned tag = (unsigned)seq;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  double v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = lat_wave_sum(s[j]);
  __syncthreads();  // (lds may still be read by the previous call)
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) lds[j * WAVES + wave] = v[j];
  }
  __syncthreads();
  if ((int)threadIdx.x < NV) {  // thread j folds and stores sum j
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += lds[threadIdx.x * WAVES + w];
    co_store_slot(slots + ((size_t)(seq & 1) * 256 + blockIdx.x) * stride + 16 * threadIdx.x, tag, t);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = 0.0;
  if (threadIdx.x < gridDim.x) {  // gridDim.x <= 256: thread t (waves 0 .. 3) watches block t
    const char *slot = slots + ((size_t)(seq & 1) * 256 + threadIdx.x) * stride;
    const long long t0 = wall_clock64();
    for (int spins = 0;; ++spins) {
      if (co_load_slot_n<NV>(slot, tag, v)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 1023) == 1023 &&
          (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = 0.0;
        break;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = lat_wave_sum(v[j]);
  __syncthreads();
  if (lane == 0 && wave < 4) {
#pragma unroll
    for (int j = 0; j < NV; ++j) lds[j * WAVES + wave] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) s[j] = (lds[j * WAVES] + lds[j * WAVES + 1]) + (lds[j * WAVES + 2] + lds[j * WAVES + 3]);
}


// The same in TWO LEVELS (round 4).  In the flat form above every block polls every block's slot: NV x 256 x 256 16-byte
// requests per polling round, each one a trip through the fabric (sc1 loads miss through the XCDs' L2s) -- measured
// 2.9 us for one value, 8.9 for three, 9.4 for six (GMRES's Gram-Schmidt chain at 128^3, the memory system otherwise
// idle): a synchronisation point was paying for its polling traffic.  Here the blocks form groups of 32 by index; a
// group's first block gathers its members' sums and publishes the group's sum; every block then polls the (at most 8)
// group sums: 256 + 8 x 256 / 32 slot reads per round instead of 65 536.  Two hops instead of one, a fixed folding order
// (lanes, then groups: the same bits in every block, run to run).  slots: [2 parities][256 blocks] of `stride` bytes, then
// [2][8] group slots of `stride` bytes.
constexpr int kCoGroup = 32;
template <int NV, int WAVES>
__device__ __forceinline__ void co_allreduce2_n(double (&s)[NV], char *slots, int stride, int *gave_up, unsigned long long seq,
                                                double *lds /* [NV * WAVES] */) {
  const unsigned tag = (unsigned)seq;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  double v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = lat_wave_sum(s[j]);
  __syncthreads();  // (lds may still be read by the previous call)
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) lds[j * WAVES + wave] = v[j];
  }
  __syncthreads();
  if (wave == 0) {
    char *group_slots = slots + (size_t)2 * 256 * stride;
    const unsigned b = blockIdx.x, g = b / kCoGroup, ng = (gridDim.x + kCoGroup - 1) / kCoGroup;
    if (lane < NV) {  // lane j folds and stores the block's sum j
      double t = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) t += lds[lane * WAVES + w];
      co_store_slot(slots + ((size_t)(seq & 1) * 256 + b) * stride + 16 * lane, tag, t);
    }
    auto poll = [&](const char *slot, bool active) {
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = 0.0;
      if (active) {
        const long long t0 = wall_clock64();
        for (int spins = 0;; ++spins) {
          if (co_load_slot_n<NV>(slot, tag, v)) break;
          __builtin_amdgcn_s_sleep(1);
          if ((spins & 1023) == 1023 &&
              (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = 0.0;
            break;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = lat_wave_sum(v[j]);  // lanes in order: a fixed tree
    };
    if (b % kCoGroup == 0) {  // the group's first block: gather the members' sums, publish the group's
      const unsigned members = min((unsigned)kCoGroup, gridDim.x - b);
      poll(slots + ((size_t)(seq & 1) * 256 + b + lane) * stride, (unsigned)lane < members);
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (lane == j) co_store_slot(group_slots + ((size_t)(seq & 1) * 8 + g) * stride + 16 * j, tag, v[j]);
    }
    poll(group_slots + ((size_t)(seq & 1) * 8 + lane) * stride, (unsigned)lane < ng);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) lds[j * WAVES] = v[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) s[j] = lds[j * WAVES];
}


// The flat all-reduce with DENSE, value-major slots (round 4): value j of block b is ONE 16-byte granule at
// ((parity * 16 + j) * 256 + b) * 16.  Every block still reads every block's values -- one hop --, but as whole
// lines: thread t polls the granules t, t + T, ... of the NV x gridDim.x in flight (all loads of a thread in one round
// trip), 16 x NV x gridDim.x contiguous bytes per block and round instead of NV x gridDim.x scattered lines.  The values
// go through LDS; value j is folded by lanes over the blocks in a fixed order (the same bits in every block, run to
// run).  lds: NV x 256 + NV doubles.  slots: 2 x 16 x 256 x 16 bytes.  T = WAVES x 64 threads, all of them poll.
constexpr int kDenseMaxValues = 16;
// (in two halves, like res_allreduce: a caller may request loads of its own between the block's arrival and its wait for
//  the others -- the Gram-Schmidt chain asks for the next group's vectors there)
template <int NV, int WAVES>
__device__ __forceinline__ void co_allreduce_dense_arrive(const double (&s)[NV], char *slots, unsigned long long seq, double *lds) {
  static_assert(NV <= kDenseMaxValues && NV <= 2 * WAVES, "co_allreduce_dense: too many values");
  constexpr int T = WAVES * kWave;
  constexpr int MAXG = (256 * NV + T - 1) / T;  // granules per thread, at most
  static_assert(MAXG <= 5, "co_allreduce_dense: at most five granules per thread");
  const unsigned tag = (unsigned)seq;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  double v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = lat_wave_sum(s[j]);
  __syncthreads();  // (lds may still be read by the previous call)
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) lds[j * WAVES + wave] = v[j];
  }
  __syncthreads();
  char *base = slots + (size_t)(seq & 1) * kDenseMaxValues * 256 * 16;
  if ((int)threadIdx.x < NV) {  // thread j folds and stores the block's sum j
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += lds[threadIdx.x * WAVES + w];
    co_store_slot(base + ((size_t)threadIdx.x * 256 + blockIdx.x) * 16, tag, t);
  }
  __syncthreads();  // (the partials in lds are consumed: the polled values take their place)
}
template <int NV, int WAVES>
__device__ __forceinline__ void co_allreduce_dense_wait(double (&s)[NV], char *slots, int *gave_up, unsigned long long seq, double *lds) {
  constexpr int T = WAVES * kWave;
  constexpr int MAXG = (256 * NV + T - 1) / T;  // granules per thread, at most
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned tag = (unsigned)seq;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int nb = gridDim.x;
  char *base = slots + (size_t)(seq & 1) * kDenseMaxValues * 256 * 16;
  // this thread's granules: index g = t + T i over [value][block 0 .. 255]; blocks >= nb do not exist
  const char *p[MAXG];
  bool need[MAXG];
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int g = (int)threadIdx.x + T * i;
    need[i] = g < 256 * NV && (g & 255) < nb;
    p[i] = base + (size_t)(need[i] ? g : 0) * 16;
  }
  u32x4 w[MAXG];
  const long long t0 = wall_clock64();
  for (int spins = 0;; ++spins) {
    if constexpr (MAXG == 1)
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w[0]) : "v"(p[0]) : "memory");
    else if constexpr (MAXG == 2)
      asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]) : "v"(p[0]), "v"(p[1]) : "memory");
    else if constexpr (MAXG == 3)
      asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]) : "v"(p[0]), "v"(p[1]), "v"(p[2]) : "memory");
    else if constexpr (MAXG == 4)
      asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                   "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
    else
      asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                   "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4])
                   : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]) : "memory");
    bool ok = true;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) ok = ok && (!need[i] || (w[i].y == tag && w[i].w == tag));
    if (ok) break;
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 1023) == 1023 &&
        (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  double *vals = lds;  // [NV][256]
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int g = (int)threadIdx.x + T * i;
    if (g < 256 * NV) vals[g] = need[i] ? __hiloint2double((int)w[i].z, (int)w[i].x) : 0.0;
  }
  __syncthreads();
  // value j by wave j % WAVES: lanes over the blocks l, l + 64, l + 128, l + 192, then the wave's tree -- a fixed order
  double *res = lds + NV * 256;
#pragma unroll
  for (int r = 0; r < (NV + WAVES - 1) / WAVES; ++r) {
    const int j = wave + r * WAVES;
    if (j < NV) {
      double t = (vals[j * 256 + lane] + vals[j * 256 + 64 + lane]) + (vals[j * 256 + 128 + lane] + vals[j * 256 + 192 + lane]);
      t = lat_wave_sum(t);
      if (lane == 0) res[j] = t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) s[j] = res[j];
}
// (the two halves in one piece, as it was before the split: composing it from them cost the S = 8 chain kernel 2 us per
//  inner iteration at 128^3 -- code placement, not instructions)
template <int NV, int WAVES>
__device__ __forceinline__ void co_allreduce_dense(double (&s)[NV], char *slots, int *gave_up, unsigned long long seq, double *lds) {
  static_assert(NV <= kDenseMaxValues && NV <= 2 * WAVES, "co_allreduce_dense: too many values");
  constexpr int T = WAVES * kWave;
  constexpr int MAXG = (256 * NV + T - 1) / T;  // granules per thread, at most
  static_assert(MAXG <= 5, "co_allreduce_dense: at most five granules per thread");
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned tag = (unsigned)seq;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int nb = gridDim.x;
  double v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = lat_wave_sum(s[j]);
  __syncthreads();  // (lds may still be read by the previous call)
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) lds[j * WAVES + wave] = v[j];
  }
  __syncthreads();
  char *base = slots + (size_t)(seq & 1) * kDenseMaxValues * 256 * 16;
  if ((int)threadIdx.x < NV) {  // thread j folds and stores the block's sum j
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += lds[threadIdx.x * WAVES + w];
    co_store_slot(base + ((size_t)threadIdx.x * 256 + blockIdx.x) * 16, tag, t);
  }
  __syncthreads();  // (the partials in lds are consumed: the polled values take their place)
  // this thread's granules: index g = t + T i over [value][block 0 .. 255]; blocks >= nb do not exist
  const char *p[MAXG];
  bool need[MAXG];
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int g = (int)threadIdx.x + T * i;
    need[i] = g < 256 * NV && (g & 255) < nb;
    p[i] = base + (size_t)(need[i] ? g : 0) * 16;
  }
  u32x4 w[MAXG];
  const long long t0 = wall_clock64();
  for (int spins = 0;; ++spins) {
    if constexpr (MAXG == 1)
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w[0]) : "v"(p[0]) : "memory");
    else if constexpr (MAXG == 2)
      asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]) : "v"(p[0]), "v"(p[1]) : "memory");
    else if constexpr (MAXG == 3)
      asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]) : "v"(p[0]), "v"(p[1]), "v"(p[2]) : "memory");
    else if constexpr (MAXG == 4)
      asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                   "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
    else
      asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                   "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4])
                   : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]) : "memory");
    bool ok = true;
#pragma unroll
    for (int i = 0; i < MAXG; ++i) ok = ok && (!need[i] || (w[i].y == tag && w[i].w == tag));
    if (ok) break;
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 1023) == 1023 &&
        (wall_clock64() - t0 > kLatTimeoutTicks || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  double *vals = lds;  // [NV][256]
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int g = (int)threadIdx.x + T * i;
    if (g < 256 * NV) vals[g] = need[i] ? __hiloint2double((int)w[i].z, (int)w[i].x) : 0.0;
  }
  __syncthreads();
  // value j by wave j % WAVES: lanes over the blocks l, l + 64, l + 128, l + 192, then the wave's tree -- a fixed order
  double *res = lds + NV * 256;
#pragma unroll
  for (int r = 0; r < (NV + WAVES - 1) / WAVES; ++r) {
    const int j = wave + r * WAVES;
    if (j < NV) {
      double t = (vals[j * 256 + lane] + vals[j * 256 + 64 + lane]) + (vals[j * 256 + 128 + lane] + vals[j * 256 + 192 + lane]);
      t = lat_wave_sum(t);
      if (lane == 0) res[j] = t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) s[j] = res[j];
}

}  // namespace storm
