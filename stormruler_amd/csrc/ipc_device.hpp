// Device side of the peer-window transport (comm.hip): the in-kernel all-reduce -- usable from any ONE-BLOCK kernel
// and, as a one-wave variant, by the block that finishes a ticketed reduction -- and the halo primitives.
//
// NOTHING that crosses ranks relies on the ORDER in which two stores become visible: every 8-byte word carries its own
// validity tag.  A double travels as two words { tag | low half }, { tag | high half } (tag = low 32 bits of the
// exchange's epoch, never 0), each ONE naturally aligned store -- single-copy atomic on the device, over xGMI and on
// PCIe -- and a reader takes a value when both words carry the tag it expects, polling with system-scope
// (cache-bypassing) loads until they do.  No flag follows the data, no release fence precedes a flag, no
// acknowledged-store assumption (csrc/ticket_device.hpp documents that an acknowledged write-through store is not yet
// visible to another XCD under load); whole-cache system fences (L2 write-back + invalidate) are never issued -- with
// the L2s full of a streaming kernel's lines they cost tens of microseconds (63 us per CG iteration, measured).
// Buffer reuse is guarded separately: the all-reduce by its double-buffered handshake (below), the halo segments by
// an acknowledgement word the receiver stores once its kernels have consumed a plane.
#pragma once

#include "common.hpp"

namespace storm {

constexpr int kIpcArVals = 64;
constexpr int64_t kIpcArSlot = 1024;  // 64 values x two self-validating 8-byte words
constexpr long long kIpcTimeoutTicks = 500000000LL;  // 5 s of the 100 MHz real-time counter
constexpr int kIpcMaxEntries = 16;    // halo-plan entries (neighbour segments) a kernel takes as arguments

// Window of rank r (offsets multiples of 256 bytes; P = n_ranks; "parity" = epoch & 1 double-buffers everything):
//   [all-reduce slots]  2 x P x kIpcArSlot   slot (parity, s): the values rank s contributed, tagged words
//   [halo acks       ]  P x 64               ack (d): last halo epoch rank d has consumed of what THIS rank sent it
//   [counters        ]  256                  all-reduce epoch (advanced by the device, only by all-reduces that run),
//                                            the boundary kernels' ticket
//   [halo data       ]  2 x P x seg_bytes    data (parity, s): the rows rank s sends here, 16 bytes per value (tagged)
struct IpcDev {
  char *const *peers;
  char *local;
  int n_ranks, rank;
  int64_t ar_off, ack_off, ctr_off, data_off, seg_bytes;
  int *error;
  long long *stat;  // where the time of the exchanges goes (storm_hip_ctx_get_counter "ipc_*"): [0] ticks of the 100 MHz counter spent
                    // waiting in all-reduces, [1] all-reduces, [2] ticks waiting for acknowledgements before a send, [3] such waits,
                    // [4] thread-ticks of halo values that were not there at the first look, [5] such values
};
__device__ __forceinline__ void ipc_stat_add(const IpcDev &w, int k, long long ticks) {
  if (w.stat) {
    __hip_atomic_fetch_add(w.stat + k, ticks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(w.stat + k + 1, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
typedef unsigned long long ipc_u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ ipc_u64x2 ipc_tagged(double v, unsigned long long tag_hi) {
  return ipc_u64x2{tag_hi | (unsigned)__double2loint(v), tag_hi | (unsigned)__double2hiint(v)};
}
__device__ __forceinline__ double ipc_untag(ipc_u64x2 w) {
  return __hiloint2double((int)(unsigned)w.y, (int)(unsigned)w.x);
}
__device__ __forceinline__ bool ipc_tag_ok(ipc_u64x2 w, unsigned long long tag_hi) {
  return (w.x & 0xffffffff00000000ull) == tag_hi && (w.y & 0xffffffff00000000ull) == tag_hi;
}
// one 16-byte write-through store at system scope (each 8-byte half single-copy atomic)
__device__ __forceinline__ void ipc_store16(void *p, ipc_u64x2 w) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(w) : "memory");  // (s_nop 1: the store-data hazard the compiler does not see inside asm, resident.hip res_store16)
}
// one 16-byte load that bypasses the caches of this device (a line of the window this XCD's L2 may hold is stale)
__device__ __forceinline__ ipc_u64x2 ipc_load16(const void *p) {
  ipc_u64x2 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  return w;
}
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
// Poll one tagged value until it carries `tag_hi`; 0.0 (and the sticky error flag) when the wait gives up.
__device__ __forceinline__ double ipc_poll_value(const void *p, unsigned long long tag_hi, int *error) {
  ipc_u64x2 w = ipc_load16(p);
  if (!ipc_tag_ok(w, tag_hi)) {
    const long long t0 = wall_clock64();
    do {
      __builtin_amdgcn_s_sleep(1);
      w = ipc_load16(p);
      if (wall_clock64() - t0 > kIpcTimeoutTicks) {
        *error = 1;
        return 0.0;
      }
    } while (!ipc_tag_ok(w, tag_hi));
  }
  return ipc_untag(w);
}
__device__ __forceinline__ unsigned long long *ipc_ar_epoch_word(const IpcDev &w) {
  return reinterpret_cast<unsigned long long *>(w.local + w.ctr_off);
}
__device__ __forceinline__ int *ipc_boundary_ticket(const IpcDev &w) {
  return reinterpret_cast<int *>(w.local + w.ctr_off + 64);
}

// One-shot all-reduce of <= 64 doubles: every rank stores its values (tagged) into slot (parity, rank) of EVERY
// window, polls its own window until every rank's words carry the epoch and adds the contributions IN RANK ORDER --
// the same bits on every rank, two traversals of the link.  The epoch lives in the window and is advanced by the
// device, by exactly the all-reduces that run: a kernel that returns early because the solve is `done` (the same
// decision on every rank -- it is made from all-reduced values) consumes none.  The double buffer needs no further
// handshake: a rank can only start epoch e + 2 after finishing e + 1, which needed every peer's e + 1 contribution,
// which a peer sends after it has read epoch e.
//
// All threads of ONE block; buf[0 .. count) in memory the block can read and write (global or LDS).
__device__ inline void ipc_allreduce_block(const IpcDev &w, double *buf, int count) {
  __shared__ unsigned long long epoch_sh;
  if (threadIdx.x == 0) epoch_sh = *ipc_ar_epoch_word(w) + 1;
  __syncthreads();  // buf is complete, the epoch is known
  const unsigned long long epoch = epoch_sh;
  const int par = (int)(epoch & 1);
  const unsigned long long tag = (epoch & 0xffffffffull) << 32;
  const int64_t my_slot = w.ar_off + ((int64_t)par * w.n_ranks + w.rank) * kIpcArSlot;
  for (int idx = threadIdx.x; idx < w.n_ranks * count; idx += blockDim.x) {
    const int q = idx / count, j = idx % count;
    ipc_store16(w.peers[q] + my_slot + 16 * j, ipc_tagged(buf[j], tag));
  }
  __syncthreads();  // buf may be overwritten below
  const long long t_wait = w.stat ? wall_clock64() : 0;
  if ((int)threadIdx.x < count) {
    double sum = 0.0;
    for (int q = 0; q < w.n_ranks; ++q)  // rank order: the same bits everywhere
      sum += ipc_poll_value(w.local + w.ar_off + ((int64_t)par * w.n_ranks + q) * kIpcArSlot + 16 * threadIdx.x, tag, w.error);
    buf[threadIdx.x] = sum;
  }
  if (threadIdx.x == 0) {
    *ipc_ar_epoch_word(w) = epoch;
    if (w.stat) ipc_stat_add(w, 0, wall_clock64() - t_wait);
  }
  __syncthreads();
}

// The same for ONE wavefront (all 64 lanes active, the rest of the block retired or idle): v[j], j < K, is this rank's
// contribution, the same in every lane; returns the sums in every lane.  Used by the block that draws the last
// ticket of an in-kernel reduction (ticket_device.hpp): the multi-rank reduction costs no launch at all.
template <int K>
__device__ inline void ipc_allreduce_wave(const IpcDev &w, double (&v)[K], int count) {
  const int lane = threadIdx.x & (kWave - 1);
  unsigned long long epoch = 0ull;
  if (lane == 0) epoch = *ipc_ar_epoch_word(w) + 1;
  epoch = __shfl(epoch, 0, kWave);
  const int par = (int)(epoch & 1);
  const unsigned long long tag = (epoch & 0xffffffffull) << 32;
  const int64_t my_slot = w.ar_off + ((int64_t)par * w.n_ranks + w.rank) * kIpcArSlot;
  for (int idx = lane; idx < w.n_ranks * count; idx += kWave) {
    const int q = idx / count, j = idx % count;
    double mine = v[0];
#pragma unroll
    for (int t = 1; t < K; ++t) mine = j == t ? v[t] : mine;
    ipc_store16(w.peers[q] + my_slot + 16 * j, ipc_tagged(mine, tag));
  }
  double sum = 0.0;
  const long long t_wait = w.stat ? wall_clock64() : 0;
  if (lane < count)
    for (int q = 0; q < w.n_ranks; ++q)
      sum += ipc_poll_value(w.local + w.ar_off + ((int64_t)par * w.n_ranks + q) * kIpcArSlot + 16 * lane, tag, w.error);
#pragma unroll
  for (int t = 0; t < K; ++t) v[t] = __shfl(sum, t, kWave);
  if (lane == 0) {
    *ipc_ar_epoch_word(w) = epoch;
    if (w.stat) ipc_stat_add(w, 0, wall_clock64() - t_wait);
  }
}

// ---- halo -------------------------------------------------------------------------------------------------------
// What a kernel needs to SEND this rank's rows (plan entry q: rows idx[ptr[q] .. ptr[q+1]) go to rank peer[q], to the
// value offset dst_off[q] of segment (parity, this rank) of that rank's window) ...
struct IpcSendPlan {
  int n_entries;
  int peer[kIpcMaxEntries];
  int ptr[kIpcMaxEntries + 1];      // into idx
  int dst_off[kIpcMaxEntries];      // values
  const int *idx;                   // owned rows to send
  int n_blocks;                     // blocks that share the work
  unsigned long long epoch[kIpcMaxEntries];  // per (this rank, peer) pair: entries towards one peer carry the same
};
// ... and to READ what the neighbours sent: halo row h (column n_rows + h) of entry q, recv_ptr[q] <= h < recv_ptr[q+1],
// is value src_off[q] + h - recv_ptr[q] of segment (parity, peer[q]) of the LOCAL window.
struct IpcRecvPlan {
  int n_entries;
  int peer[kIpcMaxEntries];
  int ptr[kIpcMaxEntries + 1];      // halo rows
  int src_off[kIpcMaxEntries];      // values
  unsigned long long epoch[kIpcMaxEntries];
  int n_peers;                      // distinct peers, for the acknowledgement ...
  int ack_peer[kIpcMaxEntries];
  unsigned long long ack_epoch[kIpcMaxEntries];
};

// Block `b` of the plan's n_blocks sending blocks (all 256 threads).  Waits (bounded) until every receiver has
// acknowledged the plane that used this parity's segment two exchanges ago, then stores its share of the rows.
// r != null: the rows sent are those of the NEW direction r + cb x, formed on the fly with the owner's expression (the
// fused CG step: the direction is not in memory yet when its boundary planes must leave).
__device__ inline void ipc_halo_send_block(const IpcDev &w, const IpcSendPlan &s, const double *__restrict__ x, int b,
                                           const double *__restrict__ r = nullptr, double cb = 0.0) {
  const long long t_ack = (w.stat && threadIdx.x == 0) ? wall_clock64() : 0;
  if (threadIdx.x < (unsigned)s.n_entries && s.epoch[threadIdx.x] > 2)
    (void)ipc_wait_ge(reinterpret_cast<const unsigned long long *>(w.local + w.ack_off + (int64_t)s.peer[threadIdx.x] * 64),
                      s.epoch[threadIdx.x] - 2, w.error);
  __syncthreads();
  if (w.stat && threadIdx.x == 0 && b == 0) ipc_stat_add(w, 2, wall_clock64() - t_ack);  // (block 0 of the sending blocks: once per exchange)
  const int total = s.ptr[s.n_entries];
  for (int i = b * kBlock + (int)threadIdx.x; i < total; i += s.n_blocks * kBlock) {
    int q = 0;
    while (q + 1 < s.n_entries && i >= s.ptr[q + 1]) ++q;
    const unsigned long long e = s.epoch[q];
    const int64_t seg = w.data_off + ((int64_t)(e & 1) * w.n_ranks + w.rank) * w.seg_bytes;
    const int row = s.idx[i];
    const double v = r ? __builtin_fma(cb, x[row], r[row]) : x[row];
    ipc_store16(w.peers[s.peer[q]] + seg + 16 * (int64_t)(s.dst_off[q] + i - s.ptr[q]), ipc_tagged(v, (e & 0xffffffffull) << 32));
  }
}
// Halo row h (0 <= h < ptr[n_entries]) as its sender stored it for this exchange; polls (bounded) until it is there.
__device__ __forceinline__ double ipc_halo_value(const IpcDev &w, const IpcRecvPlan &r, int h) {
  int q = 0;
  while (q + 1 < r.n_entries && h >= r.ptr[q + 1]) ++q;
  const unsigned long long e = r.epoch[q];
  const char *p = w.local + w.data_off + ((int64_t)(e & 1) * w.n_ranks + r.peer[q]) * w.seg_bytes +
                  16 * (int64_t)(r.src_off[q] + h - r.ptr[q]);
  return ipc_poll_value(p, (e & 0xffffffffull) << 32, w.error);
}
// Two halo rows at once (the pair a lane of the paired-row kernels gathers): both loads in flight together, one wait;
// the bounded poll only where a tag is not there yet.  h < 0 or h >= n_halo: no such row, 0.0.
__device__ __forceinline__ void ipc_halo_pair(const IpcDev &w, const IpcRecvPlan &r, int ha, int hb, int n_halo, double *va, double *vb) {
  const bool oa = ha >= 0 && ha < n_halo, ob = hb >= 0 && hb < n_halo;
  const char *pa = w.local, *pb = w.local;  // (a valid address for the lane that has nothing to read)
  unsigned long long ta = 0ull, tb = 0ull;
  if (oa) {
    int q = 0;
    while (q + 1 < r.n_entries && ha >= r.ptr[q + 1]) ++q;
    const unsigned long long e = r.epoch[q];
    pa = w.local + w.data_off + ((int64_t)(e & 1) * w.n_ranks + r.peer[q]) * w.seg_bytes + 16 * (int64_t)(r.src_off[q] + ha - r.ptr[q]);
    ta = (e & 0xffffffffull) << 32;
  }
  if (ob) {
    int q = 0;
    while (q + 1 < r.n_entries && hb >= r.ptr[q + 1]) ++q;
    const unsigned long long e = r.epoch[q];
    pb = w.local + w.data_off + ((int64_t)(e & 1) * w.n_ranks + r.peer[q]) * w.seg_bytes + 16 * (int64_t)(r.src_off[q] + hb - r.ptr[q]);
    tb = (e & 0xffffffffull) << 32;
  }
  ipc_u64x2 wa, wb;
  asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(wa), "=&v"(wb)
               : "v"(pa), "v"(pb)
               : "memory");
  const bool slow = (oa && !ipc_tag_ok(wa, ta)) || (ob && !ipc_tag_ok(wb, tb));
  const long long t_slow = (w.stat && slow) ? wall_clock64() : 0;
  *va = oa ? (ipc_tag_ok(wa, ta) ? ipc_untag(wa) : ipc_poll_value(pa, ta, w.error)) : 0.0;
  *vb = ob ? (ipc_tag_ok(wb, tb) ? ipc_untag(wb) : ipc_poll_value(pb, tb, w.error)) : 0.0;
  if (w.stat && slow) ipc_stat_add(w, 4, wall_clock64() - t_slow);
}
// Every block of a kernel that consumed halo values calls this at its end (all threads): the last block to arrive
// tells every sender that its plane has been consumed.
__device__ inline void ipc_halo_ack_last_block(const IpcDev &w, const IpcRecvPlan &r) {
  __syncthreads();  // every wave of the block has its halo values
  if (threadIdx.x != 0) return;
  int *t = ipc_boundary_ticket(w);
  if (__hip_atomic_fetch_add(t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (int)gridDim.x - 1) return;
  __hip_atomic_store(t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int q = 0; q < r.n_peers; ++q)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(w.peers[r.ack_peer[q]] + w.ack_off + (int64_t)w.rank * 64),
                       r.ack_epoch[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace storm
