// Device side of the peer-window transport (comm.hip): the in-kernel all-reduce, usable from any ONE-BLOCK kernel --
// the final pass of a reduction exchanges its sums with the other ranks itself and goes on to the scalar step, so a
// multi-rank reduction costs no extra launch (solvers.hip, krylov.hip) -- and the primitives the halo kernels use.
//
// Everything that crosses ranks is moved with RELAXED SYSTEM-SCOPE ATOMIC stores / loads: single write-through /
// miss-through accesses.  Ordering "values before tag" is the store acknowledgement (a workgroup-scope release =
// s_waitcnt); whole-cache system fences (L2 write-back + invalidate) are never issued -- with the L2s full of a
// streaming kernel's lines they cost tens of microseconds per reduction (measured: 63 us per CG iteration at one
// rank with them, see profiles/r02g_comm_path_overhead.json).
#pragma once

#include "common.hpp"

namespace storm {

constexpr int kIpcArVals = 64;
constexpr int64_t kIpcArSlot = 640;  // 64 doubles + tag, padded
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
__device__ inline void ipc_allreduce_block(const IpcDev &w, double *buf, int count, unsigned long long epoch) {
  const int par = (int)(epoch & 1);
  const int64_t my_slot = w.ar_off + ((int64_t)par * w.n_ranks + w.rank) * kIpcArSlot;
  __syncthreads();  // buf is complete
  for (int idx = threadIdx.x; idx < w.n_ranks * count; idx += blockDim.x) {
    const int q = idx / count, j = idx % count;
    sys_store(reinterpret_cast<double *>(w.peers[q] + my_slot) + j, buf[j]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's value stores are acknowledged ...
  __syncthreads();                                         // ... and every other wave's: only then the tags
  if ((int)threadIdx.x < w.n_ranks)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(w.peers[threadIdx.x] + my_slot + kIpcArVals * 8), epoch,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((int)threadIdx.x < w.n_ranks) {
    const int64_t slot = w.ar_off + ((int64_t)par * w.n_ranks + threadIdx.x) * kIpcArSlot;
    (void)ipc_wait_ge(reinterpret_cast<const unsigned long long *>(w.local + slot + kIpcArVals * 8), epoch, w.error);
  }
  __syncthreads();
  if ((int)threadIdx.x < count) {
    double sum = 0.0;
    for (int q = 0; q < w.n_ranks; ++q)  // rank order: the same bits everywhere
      sum += sys_load(reinterpret_cast<const double *>(w.local + w.ar_off + ((int64_t)par * w.n_ranks + q) * kIpcArSlot) +
                      threadIdx.x);
    buf[threadIdx.x] = sum;
  }
  __syncthreads();
}


}  // namespace storm
