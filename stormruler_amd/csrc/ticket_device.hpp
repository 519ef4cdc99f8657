// A reduction that finishes inside the kernel that produced its partial sums, for ANY grid size: two levels of
// tickets.  Blocks form groups of kTicketGroup; the last block of a group to arrive folds the group's partials, the
// last group-folder to arrive folds the groups' sums and owns the result -- it can run the scalar step of the
// solver right there, and the launch of a separate final-pass kernel (~4-5 us, dependent) disappears.
//
// Protocol (no cache-wide fence anywhere -- an agent-scope release writes back the whole L2 on this chip):
//   * a partial sum is published by lane 0 with an atomic EXCHANGE whose returned value it waits for: a returning
//     read-modify-write has been performed at the point of coherence (a plain write-through store is acknowledged
//     by the XCD's L2 before the data has left it -- under load a reader on another XCD could still miss it: seen
//     as a rare wrong sum at 256^3); only then it draws its ticket with a relaxed agent-scope atomic add; whoever
//     draws the last ticket therefore finds every partial at the coherence point and reads it with coherent loads;
//   * counters re-arm themselves (the drawer of the last ticket stores 0), so a buffer zeroed once serves forever;
//   * only wave 0 of a block takes part (the other waves retire as soon as the block's partial is formed);
//   * fixed folding order (lane i takes entry i, i + 64, ...; xor-shuffle tree): run-to-run reproducible.
#pragma once
#include "common.hpp"
#include "wave_device.hpp"

namespace storm {

constexpr int kTicketGroup = 64;
constexpr int kTicketStride = 16;          // ints between counters (one 64-byte line each)
constexpr int kTicketMaxGroups = 2048;     // 131 072 blocks
// every streaming kernel that finishes its reduction with tickets (blas1.hip, krylov.hip, solvers.hip) sizes its grid
// with stream_blocks(): whatever the row count, its blocks fit the counters and the group-sum buffer
static_assert(kMaxStreamBlocks <= kTicketGroup * kTicketMaxGroups, "stream_blocks() may exceed the ticket counters");

struct TicketArgs {
  int *cnt;       // [1 + kTicketMaxGroups] counters, kTicketStride apart; null = tickets off
  double *part1;  // [K][n_blocks] block partials
  double *part2;  // [K][n_groups] group partials
};

// Store v[0 .. k) to p[j * stride] so that they are at the point of coherence when the function returns: atomic
// EXCHANGES, all issued back to back, whose returned values are then consumed -- a returning read-modify-write has
// been performed at the point of coherence; the empty statement that consumes them is also a compiler barrier for
// memory, so nothing later is issued before they are back.
template <int KMAX>
__device__ __forceinline__ void ticket_publish(double *p, size_t stride, const double (&v)[KMAX], int k) {
  unsigned long long seen = 0ull;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < k) {
#ifdef STORM_TICKET_STORE_EXPERIMENT  // (what the first version did; kept to reproduce the failure)
      __hip_atomic_store(p + (size_t)j * stride, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      seen ^= __hip_atomic_exchange(reinterpret_cast<unsigned long long *>(p + (size_t)j * stride),
                                    (unsigned long long)__double_as_longlong(v[j]), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
  }
#ifdef STORM_TICKET_STORE_EXPERIMENT
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#endif
  asm volatile("" : : "v"(seen) : "memory");
}
__device__ __forceinline__ double ticket_wave_sum(double v) {
  return wave_sum_all(v);  // (= the xor butterfly's value in every lane, bit for bit: wave_device.hpp)
}

// Wave 0 of every block calls this (all 64 lanes) with the block's partials `mine[0 .. k)` (k <= KMAX).
// Returns true in wave 0 of exactly one block, the last to arrive, with total[j] valid in all lanes.
template <int KMAX>
__device__ __forceinline__ bool ticket_reduce_wave0(const TicketArgs &t, const double (&mine)[KMAX], int k, unsigned bx,
                                                    unsigned nb, double (&total)[KMAX]) {
  const unsigned lane = threadIdx.x & (kWave - 1);
  const unsigned g = bx / kTicketGroup, ng = (nb + kTicketGroup - 1) / kTicketGroup;
  const unsigned gsize = (nb - g * kTicketGroup) < (unsigned)kTicketGroup ? (nb - g * kTicketGroup) : (unsigned)kTicketGroup;
  int go = 0;
  if (lane == 0) {
    ticket_publish<KMAX>(t.part1 + bx, nb, mine, k);
    int *c = t.cnt + (size_t)(1 + g) * kTicketStride;
    if (__hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gsize - 1) {
      __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      go = 1;
    }
  }
  go = __shfl(go, 0, kWave);
  if (!go) return false;
  double gp[KMAX];
  // (all loads first, then the sums: a load per value waited for in turn costs a trip to memory each -- with ten values
  //  that was 12 us per group, and 24 - 57 us of serial tail in the last fold below)
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    gp[j] = 0.0;
    if (j < k && lane < gsize)
      gp[j] = __hip_atomic_load(t.part1 + (size_t)j * nb + (size_t)g * kTicketGroup + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int j = 0; j < KMAX; ++j) gp[j] = ticket_wave_sum(gp[j]);
  go = 0;
  if (lane == 0) {
    ticket_publish<KMAX>(t.part2 + g, ng, gp, k);
    if (__hip_atomic_fetch_add(t.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)ng - 1) {
      __hip_atomic_store(t.cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      go = 1;
    }
  }
  go = __shfl(go, 0, kWave);
  if (!go) return false;
  // lane i takes groups i, i + 64, ... in ascending order (the order of the sums is fixed); four groups per lane and
  // value are loaded before the first is added
  constexpr int kAhead = KMAX <= 4 ? 4 : 2;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) total[j] = 0.0;
  for (unsigned i0 = lane; i0 < ng; i0 += kWave * kAhead) {
    double v[KMAX][kAhead];
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const unsigned i = i0 + u * kWave;
        v[j][u] = 0.0;
        if (j < k && i < ng) v[j][u] = __hip_atomic_load(t.part2 + (size_t)j * ng + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
#pragma unroll
      for (int u = 0; u < kAhead; ++u)
        if (i0 + u * kWave < ng) total[j] += v[j][u];
  }
#pragma unroll
  for (int j = 0; j < KMAX; ++j) total[j] = ticket_wave_sum(total[j]);
  return true;
}

}  // namespace storm
