// Device side of the peer-window transport (comm.hip): the in-kernel all-reduce, usable from any ONE-BLOCK kernel --
// the final pass of a reduction exchanges its sums with the other ranks itself and goes on to the scalar step, so a
// multi-rank reduction costs no extra launch (solvers.hip, krylov.hip) -- and the primitives the halo kernels use.
//
// Everything that crosses ranks is moved with RELAXED SYSTEM-SCOPE ATOMIC stores / loads: single write-through /
// miss-through accesses.  The all-reduce's values validate themselves (no ordering needed); the halo planes are ordered
// before their flag by a kernel boundary; whole-cache system fences (L2 write-back + invalidate) are never issued -- with the L2s full of a
// streaming kernel's lines they cost tens of microseconds per reduction (measured: 63 us per CG iteration at one
// rank with them, see profiles/r02g_comm_path_overhead.json).
#pragma once

#include "common.hpp"

namespace storm {

constexpr int kIpcArVals = 64;
constexpr int64_t kIpcArSlot = 1024;  // 64 values x two self-validating 8-byte words
constexpr long long kIpcTimeoutTicks = 500000000LL;  // 5 s
struct IpcDev {
  char *const *peers;
  char *local;
  int n_ranks, rank;
  int64_t ar_off, flag_off, ack_off, data_off, seg_bytes;
  int *error;
};
__device__ __forceinline__ void sys_store(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double sys_load(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool ipc_wait_ge(const unsigned long long *word, unsigned long long target, int *error) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > kIpcTimeoutTicks) {
      *error = 1;
      return false;
    }
  }
  return true;
}

// All threads of ONE block; buf[0 .. count) in memory the block can read and write (global or LDS).
// A value travels as two self-validating 8-byte words { low half, tag } { high half, tag } (tag = low 32 bits of the
// epoch), each ONE atomic store: the writer needs no ordering between values and a separate tag, and no
// acknowledgement -- an acknowledged write-through store is not yet visible everywhere (ticket_device.hpp: measured
// on one device under load) -- and the reader takes a value when both of its words carry the current tag.
__device__ inline void ipc_allreduce_block(const IpcDev &w, double *buf, int count, unsigned long long epoch) {
  const int par = (int)(epoch & 1);
  const unsigned long long tag = (epoch & 0xffffffffull) << 32;
  const int64_t my_slot = w.ar_off + ((int64_t)par * w.n_ranks + w.rank) * kIpcArSlot;
  __syncthreads();  // buf is complete
  for (int idx = threadIdx.x; idx < w.n_ranks * count; idx += blockDim.x) {
    const int q = idx / count, j = idx % count;
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(w.peers[q] + my_slot) + 2 * j;
    const double v = buf[j];
    __hip_atomic_store(dst, tag | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, tag | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();  // buf may be overwritten below
  if ((int)threadIdx.x < count) {
    double sum = 0.0;
    const long long t0 = wall_clock64();
    for (int q = 0; q < w.n_ranks; ++q) {  // rank order: the same bits everywhere
      const unsigned long long *src =
          reinterpret_cast<const unsigned long long *>(w.local + w.ar_off + ((int64_t)par * w.n_ranks + q) * kIpcArSlot) +
          2 * threadIdx.x;
      unsigned long long lo, hi;
      for (;;) {
        lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((lo & 0xffffffff00000000ull) == tag && (hi & 0xffffffff00000000ull) == tag) break;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > kIpcTimeoutTicks) {
          *w.error = 1;
          break;
        }
      }
      sum += __hiloint2double((int)(unsigned)hi, (int)(unsigned)lo);
    }
    buf[threadIdx.x] = sum;
  }
  __syncthreads();
}


}  // namespace storm
