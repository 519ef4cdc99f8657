// Multi-GPU plumbing: RCCL over xGMI.
//
// One process per GPU.  Two communicators per context so the two traffic classes never
// queue behind each other: `halo` carries the point-to-point halo planes on the comm stream
// (overlapped with the interior rows of the SpMV on the compute stream), `red` carries the
// 8..400-byte dot-product all-reduces on the compute stream (latency-bound, so reductions
// are batched by the solvers: BiCGStab's (<t,r>,<t,t>) and (|r|^2,<rt,r>) are one call each).
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "ipc_device.hpp"
#include "solver_device.hpp"

namespace storm {

struct Comm {
  ncclComm_t halo = nullptr;
  ncclComm_t red = nullptr;
  // Host-staged transport (storm_hip_ctx_comm_init_host): the same protocol with the bytes carried by
  // callbacks of the host program (MPI, gloo, ...) instead of RCCL.  Synchronous; for hosts without a usable
  // RCCL and for running the multi-rank path with several ranks on ONE device (tests/test_gpu_two_ranks.py).
  storm_hip_allreduce_fn host_allreduce = nullptr;
  storm_hip_exchange_fn host_exchange = nullptr;
  void *host_user = nullptr;
  double *h_stage = nullptr;  // pinned: [send | recv] halo values, or the reduction scalars
  int64_t h_stage_len = 0;
  // Peer-window transport (storm_hip_ctx_comm_init_ipc): every rank owns one window of device memory that all
  // ranks map (hipIpc); halo planes and reduction scalars are WRITTEN INTO THE RECEIVER'S WINDOW by the sender's
  // kernels and picked up by polling -- no RCCL kernel, no staging protocol.  See the IPC section below.
  bool ipc = false;
  char *win_local = nullptr;            // this rank's window (hipMalloc)
  std::vector<char *> win_peer;         // [n_ranks] mapped windows (win_peer[rank] == win_local)
  char **d_win_peer = nullptr;          // the same on the device
  int64_t win_bytes = 0, seg_bytes = 0;
  std::vector<unsigned long long> pair_epoch;  // [n_ranks] halo exchanges this rank has had with each peer (both sides count alike)
  int *d_error = nullptr, *h_error = nullptr;  // set by a kernel whose wait timed out
  long long *d_stat = nullptr;          // ipc_device.hpp IpcDev::stat (8 counters, zeroed at init)
  double *pending_x = nullptr;          // generic form of the exchange: the receive half runs in comm_halo_exchange_end
  const double *prebegun = nullptr;     // RCCL: the exchange of THIS vector's halo is in flight already (comm_halo_exchange_begin_formed)
  IpcRecvPlan pending_recv;
  // RCCL transport, option profile_comm: where an exchange and an all-reduce spend their time, from the device's own clock.
  // One-thread stamp kernels between the launches of BOTH streams store wall_clock64() (ticks of 10 ns) into these rings --
  // an instrumented solve, run beside the timed one (every stamp is a launch of its own, ~2 us on its stream).
  // RCCL transport, option rccl_flag_wait: the boundary rows are released by a FLAG in device memory instead of a
  // cross-stream event -- a one-thread kernel behind the send / recv group on the comm stream stores the exchange's number,
  // a one-thread kernel in front of the boundary launch on the compute stream polls for it (bounded).  The event costs ~13 us
  // between "exchange done" and "boundary rows start" (profile_comm: halo_done_to_boundary_rows); the flag ~3 us.
  unsigned long long *d_flag = nullptr;  // [0]: number of the last completed exchange
  unsigned long long flag_seq = 0;       // exchanges begun
  unsigned long long ready_seq = 0;      // [8]: hand-offs compute stream -> comm stream
  long long *d_prof = nullptr;          // [kProfRing][8] exchanges: A, P, Q, R, B, C, -, - ; then [kProfRing][2] all-reduces: E, F
  long long prof_ex = 0, prof_ar = 0;   // exchanges / all-reduces stamped since profile_comm was set
  std::vector<unsigned char> prof_mode; // per exchange: 0 = packed on the comm stream, 1 = packed (formed) on the compute stream
};
constexpr int kProfRing = 8192;
__global__ void comm_stamp_kernel(long long *slot) { *slot = wall_clock64(); }
__global__ void comm_flag_set_kernel(unsigned long long *flag, unsigned long long seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// (the receive kernel of the exchange ended before the setter started, on the comm stream: what it wrote is visible to
//  the kernel launched behind this one on the compute stream)
__global__ void comm_flag_wait_kernel(const unsigned long long *flag, unsigned long long seq, int *error, long long limit) {
  const long long t0 = wall_clock64();
  // `limit` ticks of 10 ns (option comm_wait_seconds: 120 s; the first exchanges of a communicator at least 180 s -- RCCL
  // sets its point-to-point connections up inside the first send / recv of every pair): the peer never came.  What runs
  // behind this kernel then reads a stale halo -- the error word makes the solve's next checked call, at the latest its
  // end, return STORM_HIP_E_COMM instead of that result.
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < seq) {
    if (wall_clock64() - t0 > limit) {
      if (error) __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    // (short naps while the hand-off is a matter of microseconds; long ones once it clearly is not)
    if (++spins < 4096) __builtin_amdgcn_s_sleep(2);
    else __builtin_amdgcn_s_sleep(127);
  }
}
static inline long long flag_wait_limit(const storm_hip_ctx *c, unsigned long long seq) {
  const long long s = seq <= 4 ? std::max<long long>(c->opt_comm_wait_seconds, 180) : c->opt_comm_wait_seconds;
  return s * 100000000ll;
}
static inline bool flag_on(const storm_hip_ctx *c) { return c->opt_rccl_flag_wait != 0 && c->comm->d_flag != nullptr; }
// "the vector is ready" from the compute stream to the comm stream: a flag too (a recorded event is a barrier with a
// system-scope release between two kernels of the compute stream, ~6 us; the one-thread setter ~2).  Both kernels are
// submitted in this order, so the waiter can never sit in front of its setter in a shared hardware queue.
static inline int ready_handoff(storm_hip_ctx *c) {
  if (flag_on(c)) {
    const unsigned long long seq = ++c->comm->ready_seq;
    hipLaunchKernelGGL(comm_flag_set_kernel, dim3(1), dim3(1), 0, c->stream, c->comm->d_flag + 8, seq);
    hipLaunchKernelGGL(comm_flag_wait_kernel, dim3(1), dim3(1), 0, c->comm_stream, c->comm->d_flag + 8, seq, c->comm->d_error,
                       flag_wait_limit(c, seq));
    HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
  }
  HIP_TRY(hipEventRecord(c->ev_x_ready, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_x_ready, 0));
  return STORM_HIP_OK;
}
// behind the send / recv group of an exchange, on the comm stream
static inline void flag_publish(storm_hip_ctx *c) {
  ++c->comm->flag_seq;
  if (flag_on(c)) hipLaunchKernelGGL(comm_flag_set_kernel, dim3(1), dim3(1), 0, c->comm_stream, c->comm->d_flag, c->comm->flag_seq);
}
static inline void prof_stamp(storm_hip_ctx *c, hipStream_t st, long long idx, int k) {
  if (idx < 0 || idx >= kProfRing) return;
  hipLaunchKernelGGL(comm_stamp_kernel, dim3(1), dim3(1), 0, st, c->comm->d_prof + idx * 8 + k);
}
static inline bool prof_on(const storm_hip_ctx *c) { return c->opt_profile_comm != 0 && c->comm != nullptr && c->comm->d_prof != nullptr; }

static int host_stage(storm_hip_ctx *c, int64_t len) {
  Comm *cm = c->comm;
  if (len <= cm->h_stage_len) return STORM_HIP_OK;
  if (cm->h_stage) (void)hipHostFree(cm->h_stage);
  cm->h_stage = nullptr, cm->h_stage_len = 0;
  HIP_TRY(hipHostMalloc((void **)&cm->h_stage, sizeof(double) * (size_t)len, hipHostMallocDefault));
  cm->h_stage_len = len;
  return STORM_HIP_OK;
}

#define NCCL_TRY(expr)                                                                          \
  do {                                                                                          \
    ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess)                                                                      \
      STORM_FAIL(STORM_HIP_E_COMM, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, \
                 __LINE__);                                                                     \
  } while (0)

// ---- peer-window transport (protocol and window layout: ipc_device.hpp) -------------------------------------------
__global__ __launch_bounds__(kBlock) void ipc_allreduce_kernel(IpcDev w, double *buf, int count) {
  ipc_allreduce_block(w, buf, count);
}
// Stand-alone halves of a halo exchange, for SpMV kernels that carry neither (spmv.hip fuses the send into the
// interior launch and the receive into the boundary launch where it can):
__global__ __launch_bounds__(kBlock) void ipc_halo_send_kernel(IpcDev w, IpcSendPlan s, const double *__restrict__ x) {
  ipc_halo_send_block(w, s, x, (int)blockIdx.x);
}
// ... every halo row behind the owned rows of x (where kernels without a window reader look for it), then the
// acknowledgement by the last block.
__global__ __launch_bounds__(kBlock) void ipc_halo_recv_copy_kernel(IpcDev w, IpcRecvPlan r, double *__restrict__ x_halo) {
  const int total = r.ptr[r.n_entries];
  for (int h = (int)(blockIdx.x * kBlock + threadIdx.x); h < total; h += (int)gridDim.x * kBlock)
    x_halo[h] = ipc_halo_value(w, r, h);
  ipc_halo_ack_last_block(w, r);
}

static IpcDev ipc_dev(const storm_hip_ctx *c) {
  const Comm *cm = c->comm;
  const int64_t P = c->n_ranks;
  IpcDev w;
  w.peers = cm->d_win_peer, w.local = cm->win_local, w.n_ranks = c->n_ranks, w.rank = c->rank;
  w.ar_off = 0;
  w.ack_off = 2 * P * kIpcArSlot;
  w.ctr_off = (w.ack_off + P * 64 + 255) / 256 * 256;
  w.data_off = w.ctr_off + 256;
  w.seg_bytes = cm->seg_bytes;
  w.error = cm->d_error;
  w.stat = cm->d_stat;
  return w;
}
static int64_t ipc_header_bytes(int64_t P) { return (2 * P * kIpcArSlot + P * 64 + 255) / 256 * 256 + 256; }
static int ipc_check_error(storm_hip_ctx *c) {
  if (c->comm && !c->comm->ipc && c->comm->h_error != nullptr && *(volatile int *)c->comm->h_error != 0) {
    // Reported ONCE, to the call that suffered it: the sequence numbers of the flags only grow, so the late exchange still
    // completes its own number and the next solve starts clean -- one skewed rank does not poison the context.
    *(volatile int *)c->comm->h_error = 0;
    STORM_FAIL(STORM_HIP_E_COMM,
               "RCCL transport: a stream hand-off (halo exchange done / vector ready) was not seen within %lld s (option "
               "comm_wait_seconds; rank %d of %d): the result of the call in flight is not valid",
               (long long)c->opt_comm_wait_seconds, c->rank, c->n_ranks);
  }
  if (c->comm && c->comm->ipc && *(volatile int *)c->comm->h_error != 0)
    STORM_FAIL(STORM_HIP_E_COMM, "peer-window transport: a wait for another rank timed out (rank %d of %d)", c->rank,
               c->n_ranks);
  return STORM_HIP_OK;
}
// The device plans of one exchange of `op` (entry offsets inside a (sender -> receiver) segment: entries towards one
// peer follow each other, 2-value aligned).  advance: count this exchange (once per distinct peer).
static void ipc_plans(const storm_hip_op *op, bool advance, IpcSendPlan *sp, IpcRecvPlan *rp) {
  storm_hip_ctx *c = op->ctx;
  Comm *cm = c->comm;
  const HaloPlan &h = op->halo;
  sp->n_entries = rp->n_entries = h.n_nbrs;
  sp->idx = h.d_send_idx;
  rp->n_peers = 0;
  sp->ptr[0] = rp->ptr[0] = 0;
  for (int q = 0; q < h.n_nbrs; ++q) {
    const int peer = h.nbr_rank[q];
    sp->peer[q] = rp->peer[q] = peer;
    sp->ptr[q + 1] = (int)h.send_ptr[q + 1], rp->ptr[q + 1] = (int)h.recv_ptr[q + 1];
    int so = 0, ro = 0;
    bool seen = false;
    for (int q2 = 0; q2 < q; ++q2)
      if (h.nbr_rank[q2] == peer) {
        seen = true;
        so += (int)((h.send_ptr[q2 + 1] - h.send_ptr[q2] + 1) & ~(int64_t)1);
        ro += (int)((h.recv_ptr[q2 + 1] - h.recv_ptr[q2] + 1) & ~(int64_t)1);
      }
    sp->dst_off[q] = so, rp->src_off[q] = ro;
    if (!seen) {
      if (advance) ++cm->pair_epoch[(size_t)peer];
      rp->ack_peer[rp->n_peers] = peer, rp->ack_epoch[rp->n_peers] = cm->pair_epoch[(size_t)peer];
      ++rp->n_peers;
    }
    sp->epoch[q] = rp->epoch[q] = cm->pair_epoch[(size_t)peer];
  }
  const int64_t per_block = (int64_t)kBlock * 8;
  sp->n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(128, (h.n_send + per_block - 1) / per_block));
}

int comm_allreduce_sum(storm_hip_ctx *c, double *d_buf, int count) {
  if (c->comm == nullptr) return STORM_HIP_OK;
  if (c->comm->host_allreduce) {
    STORM_TRY(host_stage(c, count));
    double *h = c->comm->h_stage;
    HIP_TRY(hipMemcpyAsync(h, d_buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int rc = c->comm->host_allreduce(c->comm->host_user, h, count);
    if (rc != 0) STORM_FAIL(STORM_HIP_E_COMM, "host all-reduce callback returned %d", rc);
    HIP_TRY(hipMemcpyAsync(d_buf, h, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // h is reused by the next call
    return STORM_HIP_OK;
  }
  if (c->comm->ipc) {
    STORM_REQUIRE(count >= 1 && count <= kIpcArVals, "all-reduce of %d scalars (the peer-window slots hold %d)", count,
                  kIpcArVals);
    STORM_TRY(ipc_check_error(c));
    hipLaunchKernelGGL(ipc_allreduce_kernel, dim3(1), dim3(kBlock), 0, c->stream, ipc_dev(c), d_buf, count);
    HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
  }
  STORM_REQUIRE(c->comm && c->comm->red, "all-reduce without an initialised communicator");
  const bool prof = prof_on(c) && c->comm->prof_ar < kProfRing;
  long long *ar = prof ? c->comm->d_prof + (size_t)kProfRing * 8 + c->comm->prof_ar * 2 : nullptr;
  if (prof) hipLaunchKernelGGL(comm_stamp_kernel, dim3(1), dim3(1), 0, c->stream, ar);
  NCCL_TRY(ncclAllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, c->comm->red, c->stream));
  if (prof) {
    hipLaunchKernelGGL(comm_stamp_kernel, dim3(1), dim3(1), 0, c->stream, ar + 1);
    ++c->comm->prof_ar;
  }
  return STORM_HIP_OK;
}

__global__ __launch_bounds__(kBlock) void halo_pack_kernel(int64_t n, const int *__restrict__ idx,
                                                           const double *__restrict__ x,
                                                           double *__restrict__ buf) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) buf[i] = x[idx[i]];
}

// The fused CG step over RCCL: the rows to send are those of the NEW direction p' = r + cb p, which is not in memory
// yet when its boundary planes must leave -- formed here with the owner's expression (cg_xp_kernel's, the marching
// kernel's: the same bits), cb from the device slab.
__global__ __launch_bounds__(kBlock) void halo_pack_direction_kernel(int64_t n, const int *__restrict__ idx,
                                                                     const double *__restrict__ p, const double *__restrict__ r,
                                                                     const double *__restrict__ cb, double *__restrict__ buf) {
  const double beta = *cb;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) buf[i] = __builtin_fma(beta, p[idx[i]], r[idx[i]]);
}

// One exchange of op's halo on the peer-window transport: the window view and the device plans, with the pair epochs
// advanced.  The caller enqueues the send (comm_ipc_send, or IpcSendPlan handed to a kernel that sends itself) and the
// receive (comm_ipc_recv_copy, or IpcRecvPlan handed to the kernel that reads the window and acknowledges).
int comm_ipc_exchange(const storm_hip_op *op, IpcDev *w, IpcSendPlan *sp, IpcRecvPlan *rp) {
  storm_hip_ctx *c = op->ctx;
  STORM_REQUIRE(c->comm && c->comm->ipc, "peer-window exchange without the transport");
  STORM_TRY(ipc_check_error(c));
  *w = ipc_dev(c);
  ipc_plans(op, true, sp, rp);
  return STORM_HIP_OK;
}
int comm_ipc_send(const storm_hip_op *op, const double *x, const IpcDev &w, const IpcSendPlan &sp) {
  if (sp.ptr[sp.n_entries] <= 0) return STORM_HIP_OK;
  hipLaunchKernelGGL(ipc_halo_send_kernel, dim3(sp.n_blocks), dim3(kBlock), 0, op->ctx->stream, w, sp, x);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}
int comm_ipc_recv_copy(const storm_hip_op *op, double *x, const IpcDev &w, const IpcRecvPlan &rp) {
  const int total = rp.ptr[rp.n_entries];
  const int nb = std::max(1, std::min(128, (total + kBlock * 2 - 1) / (kBlock * 2)));
  hipLaunchKernelGGL(ipc_halo_recv_copy_kernel, dim3(nb), dim3(kBlock), 0, op->ctx->stream, w, rp, x + op->n_rows);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}
bool comm_is_ipc(const storm_hip_ctx *c) { return c->comm != nullptr && c->comm->ipc; }
// The peer-window transport's wait counters (IpcDev::stat), k in [0, 8); -1 when there is no such transport.
long long comm_ipc_stat(storm_hip_ctx *c, int k) {
  if (c->comm == nullptr || !c->comm->ipc || c->comm->d_stat == nullptr || k < 0 || k >= 8) return -1;
  long long v = 0;
  (void)hipStreamSynchronize(c->stream);
  if (hipMemcpy(&v, c->comm->d_stat + k, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return v;
}

int comm_halo_exchange_begin(const storm_hip_op *op, double *x) {
  storm_hip_ctx *c = op->ctx;
  const HaloPlan &h = op->halo;
  if (h.n_nbrs == 0 || c->comm == nullptr) return STORM_HIP_OK;
  if (c->comm->host_exchange) {
    // host-staged: pack on the compute stream, copy out, let the host program move the bytes, copy the
    // received planes into x's halo tail; nothing overlaps, the interior launch simply follows
    const int64_t n_recv = h.recv_ptr[h.n_nbrs];
    STORM_TRY(host_stage(c, h.n_send + n_recv));
    double *hs = c->comm->h_stage, *hr = hs + h.n_send;
    if (h.n_send > 0) {
      const int64_t need = (h.n_send + kBlock - 1) / kBlock;
      hipLaunchKernelGGL(halo_pack_kernel, dim3((int)(need > 1024 ? 1024 : need)), dim3(kBlock), 0, c->stream, h.n_send,
                         h.d_send_idx, x, h.d_sendbuf);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(hs, h.d_sendbuf, sizeof(double) * (size_t)h.n_send, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int rc = c->comm->host_exchange(c->comm->host_user, h.n_nbrs, h.nbr_rank.data(), h.send_ptr.data(), hs,
                                          h.recv_ptr.data(), hr);
    if (rc != 0) STORM_FAIL(STORM_HIP_E_COMM, "host halo-exchange callback returned %d", rc);
    if (n_recv > 0)
      HIP_TRY(hipMemcpyAsync(x + op->n_rows, hr, sizeof(double) * (size_t)n_recv, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // the staging buffer is reused by the next exchange
    return STORM_HIP_OK;
  }
  if (c->comm->ipc) {
    // Generic form: send, then -- behind the interior rows, comm_halo_exchange_end -- copy every halo row into x's
    // tail; all on the compute stream (no cross-stream events: they cost more than these kernels, profiles/r02y).
    // spmv.hip takes the fused form (comm_ipc_exchange below) where its kernels can send / read the window themselves.
    IpcDev w;
    IpcSendPlan sp;
    IpcRecvPlan rp;
    STORM_TRY(comm_ipc_exchange(op, &w, &sp, &rp));
    STORM_TRY(comm_ipc_send(op, x, w, sp));
    c->comm->pending_x = x, c->comm->pending_recv = rp;
    return STORM_HIP_OK;
  }
  STORM_REQUIRE(c->comm && c->comm->halo, "halo exchange without an initialised communicator");
  if (c->comm->prebegun != nullptr) {  // begun by comm_halo_exchange_begin_formed, for this very vector
    // (another vector's: the last enqueued iteration of a solve sent a halo nobody asked for -- this exchange queues behind
    //  it on the comm stream)
    const bool mine = c->comm->prebegun == x;
    c->comm->prebegun = nullptr;
    if (mine) return STORM_HIP_OK;
  }
  // x must be complete before it is packed
  const long long pe = prof_on(c) ? c->comm->prof_ex++ : -1;
  if (pe >= 0) c->comm->prof_mode.push_back(0), prof_stamp(c, c->stream, pe, 0);
  STORM_TRY(ready_handoff(c));
  if (pe >= 0) prof_stamp(c, c->comm_stream, pe, 1);
  if (h.n_send > 0) {
    const int64_t need = (h.n_send + kBlock - 1) / kBlock;
    const int nb = (int)(need > 1024 ? 1024 : need);
    hipLaunchKernelGGL(halo_pack_kernel, dim3(nb), dim3(kBlock), 0, c->comm_stream, h.n_send, h.d_send_idx, x,
                       h.d_sendbuf);
    HIP_TRY(hipGetLastError());
  }
  if (pe >= 0) prof_stamp(c, c->comm_stream, pe, 2);
  NCCL_TRY(ncclGroupStart());
  for (int q = 0; q < h.n_nbrs; ++q) {
    const int64_t ns = h.send_ptr[q + 1] - h.send_ptr[q], nr = h.recv_ptr[q + 1] - h.recv_ptr[q];
    if (ns > 0)
      NCCL_TRY(ncclSend(h.d_sendbuf + h.send_ptr[q], (size_t)ns, ncclDouble, h.nbr_rank[q], c->comm->halo,
                        c->comm_stream));
    if (nr > 0)
      NCCL_TRY(ncclRecv(x + op->n_rows + h.recv_ptr[q], (size_t)nr, ncclDouble, h.nbr_rank[q], c->comm->halo,
                        c->comm_stream));
  }
  NCCL_TRY(ncclGroupEnd());
  if (pe >= 0) prof_stamp(c, c->comm_stream, pe, 3);
  flag_publish(c);
  HIP_TRY(hipEventRecord(c->ev_halo_done, c->comm_stream));
  return STORM_HIP_OK;
}

bool comm_is_rccl(const storm_hip_ctx *c) {
  return c->comm != nullptr && !c->comm->ipc && c->comm->host_exchange == nullptr && c->comm->halo != nullptr;
}

// The exchange of the fused CG step on the RCCL transport: pack p' = r + cb p of the send rows on the comm stream, send,
// receive into p_out's halo tail -- while the marching launch forms p' on the owned rows and applies the operator to the
// interior planes on the compute stream; comm_halo_exchange_end makes the boundary launch wait for it.
int comm_halo_exchange_begin_direction(const storm_hip_op *op, const double *p, const double *r, const double *cb, double *p_out) {
  storm_hip_ctx *c = op->ctx;
  const HaloPlan &h = op->halo;
  STORM_REQUIRE(comm_is_rccl(c) && h.n_nbrs > 0, "fused exchange: needs the RCCL transport and a halo plan");
  const long long pe = prof_on(c) ? c->comm->prof_ex++ : -1;
  if (pe >= 0) c->comm->prof_mode.push_back(0), prof_stamp(c, c->stream, pe, 0);
  STORM_TRY(ready_handoff(c));  // beta of the ending iteration is in the slab, r and p are complete
  if (pe >= 0) prof_stamp(c, c->comm_stream, pe, 1);
  if (h.n_send > 0) {
    const int64_t need = (h.n_send + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(halo_pack_direction_kernel, dim3((int)(need > 1024 ? 1024 : need)), dim3(kBlock), 0, c->comm_stream, h.n_send,
                       h.d_send_idx, p, r, cb, h.d_sendbuf);
    HIP_TRY(hipGetLastError());
  }
  if (pe >= 0) prof_stamp(c, c->comm_stream, pe, 2);
  NCCL_TRY(ncclGroupStart());
  for (int q = 0; q < h.n_nbrs; ++q) {
    const int64_t ns = h.send_ptr[q + 1] - h.send_ptr[q], nr = h.recv_ptr[q + 1] - h.recv_ptr[q];
    if (ns > 0)
      NCCL_TRY(ncclSend(h.d_sendbuf + h.send_ptr[q], (size_t)ns, ncclDouble, h.nbr_rank[q], c->comm->halo, c->comm_stream));
    if (nr > 0)
      NCCL_TRY(ncclRecv(p_out + op->n_rows + h.recv_ptr[q], (size_t)nr, ncclDouble, h.nbr_rank[q], c->comm->halo, c->comm_stream));
  }
  NCCL_TRY(ncclGroupEnd());
  if (pe >= 0) prof_stamp(c, c->comm_stream, pe, 3);
  flag_publish(c);
  HIP_TRY(hipEventRecord(c->ev_halo_done, c->comm_stream));
  return STORM_HIP_OK;
}

// BiCGStab (and the kernel-per-statement CG) over RCCL: the halo of the vector that the NEXT update kernel will form -- s = r - alpha v
// (MODE 0, in r's place), p' = r + beta (p - omega v) (MODE 1, in p's place), CG's p' = r + beta p (MODE 2) -- leaves before that kernel runs: the rows to send are formed here with
// the owner's expression (bicg_update_kernel<false>'s, BicgPF's: the same bits) from the operands as they are NOW, on the
// COMPUTE stream (the update overwrites an operand in place), and travel on the comm stream under the update and the
// interior rows of the apply that follows.  comm_halo_exchange_begin finds the exchange begun (prebegun) and returns.
template <int MODE>
__global__ __launch_bounds__(kBlock) void halo_pack_bicg_kernel(int64_t n, const int *__restrict__ idx, const double *__restrict__ r,
                                                                const double *__restrict__ p, const double *__restrict__ v,
                                                                const double *__restrict__ sa, const double *__restrict__ sb,
                                                                double *__restrict__ buf, double *alpha_seen) {
  // MODE 0: alpha (sb given: rho and <rt, v> -- alpha is formed here, with the update kernel's expression); MODE 1: beta, omega
  const double a = (MODE == 0 && sb) ? safe_divide(*sa, *sb) : *sa, b = sb ? *sb : 0.0;
  // (option ticket_verify: the alpha the rows that LEAVE were formed with, for bicg_update_kernel to compare with its own)
  if (MODE == 0 && alpha_seen != nullptr && blockIdx.x == 0 && threadIdx.x == 0) alpha_seen[0] = a, alpha_seen[1] = 1.0;  // (value, armed)
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const int j = idx[i];
    buf[i] = MODE == 0 ? __builtin_fma(-a, v[j], r[j]) : MODE == 1 ? __builtin_fma(a, __builtin_fma(-b, v[j], p[j]), r[j]) : __builtin_fma(a, p[j], r[j]);
  }
}
int comm_halo_exchange_begin_formed(const storm_hip_op *op, int mode, const double *r, const double *p, const double *v,
                                    const double *sa, const double *sb, double *target, double *alpha_seen) {
  storm_hip_ctx *c = op->ctx;
  const HaloPlan &h = op->halo;
  STORM_REQUIRE(comm_is_rccl(c) && h.n_nbrs > 0, "formed exchange: needs the RCCL transport and a halo plan");
  const long long pe = prof_on(c) ? c->comm->prof_ex++ : -1;
  if (pe >= 0) c->comm->prof_mode.push_back(1), prof_stamp(c, c->stream, pe, 0);
  if (h.n_send > 0) {
    const int64_t need = (h.n_send + kBlock - 1) / kBlock;
    const dim3 grid((int)(need > 1024 ? 1024 : need));
    double *no_alpha = nullptr;
    if (mode == 0) hipLaunchKernelGGL(halo_pack_bicg_kernel<0>, grid, dim3(kBlock), 0, c->stream, h.n_send, h.d_send_idx, r, p, v, sa, sb, h.d_sendbuf, alpha_seen);
    else if (mode == 1) hipLaunchKernelGGL(halo_pack_bicg_kernel<1>, grid, dim3(kBlock), 0, c->stream, h.n_send, h.d_send_idx, r, p, v, sa, sb, h.d_sendbuf, no_alpha);
    else hipLaunchKernelGGL(halo_pack_bicg_kernel<2>, grid, dim3(kBlock), 0, c->stream, h.n_send, h.d_send_idx, r, p, v, sa, sb, h.d_sendbuf, no_alpha);
    HIP_TRY(hipGetLastError());
  }
  if (pe >= 0) prof_stamp(c, c->stream, pe, 2);
  // (the forming kernel's last block setting the flag itself was measured: +20 us per BiCGStab iteration -- an agent-scope
  //  release per block writes its XCD's L2 back each time; the kernel boundary in front of the one-thread setter does it once)
  STORM_TRY(ready_handoff(c));  // the rows to send are packed
  if (pe >= 0) prof_stamp(c, c->comm_stream, pe, 1);
  NCCL_TRY(ncclGroupStart());
  for (int q = 0; q < h.n_nbrs; ++q) {
    const int64_t ns = h.send_ptr[q + 1] - h.send_ptr[q], nr = h.recv_ptr[q + 1] - h.recv_ptr[q];
    if (ns > 0) NCCL_TRY(ncclSend(h.d_sendbuf + h.send_ptr[q], (size_t)ns, ncclDouble, h.nbr_rank[q], c->comm->halo, c->comm_stream));
    if (nr > 0) NCCL_TRY(ncclRecv(target + op->n_rows + h.recv_ptr[q], (size_t)nr, ncclDouble, h.nbr_rank[q], c->comm->halo, c->comm_stream));
  }
  NCCL_TRY(ncclGroupEnd());
  if (pe >= 0) prof_stamp(c, c->comm_stream, pe, 3);
  flag_publish(c);
  HIP_TRY(hipEventRecord(c->ev_halo_done, c->comm_stream));
  c->comm->prebegun = target;
  return STORM_HIP_OK;
}

// A solve is over (or begins): no exchange begun ahead of its vector may outlive it -- the vector goes back to the pool and
// another one may get its address.
void comm_forget_prebegun(storm_hip_ctx *c) {
  if (c->comm == nullptr) return;
  // An exchange begun ahead that nobody consumed (the solve's last iteration) still has a receive in flight on the comm
  // stream that writes the halo tail of that vector: whatever takes the storage next on the compute stream waits for it.
  if (c->comm->prebegun != nullptr) (void)hipStreamWaitEvent(c->stream, c->ev_halo_done, 0);
  c->comm->prebegun = nullptr;
}

int comm_halo_exchange_end(const storm_hip_op *op) {
  storm_hip_ctx *c = op->ctx;
  if (op->halo.n_nbrs == 0 || c->comm == nullptr || c->comm->host_exchange) return STORM_HIP_OK;
  if (c->comm->ipc) {  // the receive half, behind the interior rows
    if (c->comm->pending_x == nullptr) return STORM_HIP_OK;
    double *x = c->comm->pending_x;
    c->comm->pending_x = nullptr;
    return comm_ipc_recv_copy(op, x, ipc_dev(c), c->comm->pending_recv);
  }
  const long long pe = prof_on(c) ? c->comm->prof_ex - 1 : -1;  // (the exchange begun last is the one this launch waits for)
  if (pe >= 0) prof_stamp(c, c->stream, pe, 4);
  if (flag_on(c)) {
    hipLaunchKernelGGL(comm_flag_wait_kernel, dim3(1), dim3(1), 0, c->stream, c->comm->d_flag, c->comm->flag_seq, c->comm->d_error,
                       flag_wait_limit(c, c->comm->flag_seq));
    HIP_TRY(hipGetLastError());
  } else {
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_halo_done, 0));
  }
  if (pe >= 0) prof_stamp(c, c->stream, pe, 5);
  return STORM_HIP_OK;
}

// Option profile_comm: (re)start the RCCL-path profile / read it.  Sums of ticks (10 ns) over the stamped exchanges and
// all-reduces: k = 0 exchanges, 1 cross-stream event in front of the exchange (compute stream ready -> comm stream runs),
// 2 pack, 3 send/recv, 4 the halo still in flight when the interior rows had ended (the un-hidden part), 5 both done ->
// the compute stream resumes (the second cross-stream event), 6 all-reduces, 7 all-reduce ticks (stamp to stamp).
int comm_profile_reset(storm_hip_ctx *c) {
  if (c->comm == nullptr || !comm_is_rccl(c)) return STORM_HIP_OK;
  const size_t bytes = sizeof(long long) * ((size_t)kProfRing * 8 + (size_t)kProfRing * 2);
  if (c->comm->d_prof == nullptr) HIP_TRY(hipMalloc((void **)&c->comm->d_prof, bytes));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipStreamSynchronize(c->comm_stream));
  HIP_TRY(hipMemset(c->comm->d_prof, 0, bytes));
  c->comm->prof_ex = c->comm->prof_ar = 0;
  c->comm->prof_mode.clear();
  return STORM_HIP_OK;
}
long long comm_profile_read(storm_hip_ctx *c, int k) {
  if (c->comm == nullptr || c->comm->d_prof == nullptr || k < 0 || k > 7) return -1;
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->comm_stream);
  const long long n_ex = std::min<long long>(c->comm->prof_ex, kProfRing), n_ar = std::min<long long>(c->comm->prof_ar, kProfRing);
  if (k == 0) return n_ex;
  if (k == 6) return n_ar;
  std::vector<long long> h((size_t)kProfRing * 10);
  if (hipMemcpy(h.data(), c->comm->d_prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  long long sum = 0;
  if (k == 7) {
    for (long long i = 0; i < n_ar; ++i) sum += h[(size_t)kProfRing * 8 + (size_t)i * 2 + 1] - h[(size_t)kProfRing * 8 + (size_t)i * 2];
    return sum;
  }
  for (long long i = 0; i < n_ex; ++i) {
    const long long *e = h.data() + (size_t)i * 8;
    const long long A = e[0], P = e[1], Q = e[2], R = e[3], B = e[4], C = e[5];
    const bool formed = c->comm->prof_mode[(size_t)i] != 0;
    if (A == 0 || P == 0 || Q == 0 || R == 0) continue;
    if (k == 1) sum += formed ? P - Q : P - A;
    else if (k == 2) sum += formed ? Q - A : Q - P;
    else if (k == 3) sum += formed ? R - P : R - Q;
    else if (B != 0 && C != 0) {
      if (k == 4) sum += std::max<long long>(0, R - B);
      else if (k == 5) sum += C - std::max(R, B);
    }
  }
  return sum;
}


// A plan whose send count towards a neighbour differs from what that neighbour expects to receive hangs the first
// exchange inside RCCL.  Once, when the plan is set: every rank tells each neighbour how many rows it will send,
// and compares what it is told with its own receive counts.  (Same transport as the halo itself.)
int halo_plan_cross_check(const storm_hip_op *op) {
  storm_hip_ctx *c = op->ctx;
  const HaloPlan &h = op->halo;
  if (h.n_nbrs == 0 || c->comm == nullptr) return STORM_HIP_OK;
  if (c->comm->ipc) {  // every (sender -> receiver) pair owns one segment of the receiver's window: 16 bytes per value
    STORM_REQUIRE(h.n_nbrs <= kIpcMaxEntries, "op_set_halo: %d plan entries (the peer-window transport takes %d)", h.n_nbrs,
                  kIpcMaxEntries);
    IpcSendPlan sp;
    IpcRecvPlan rp;
    ipc_plans(op, false, &sp, &rp);
    for (int q = 0; q < h.n_nbrs; ++q) {
      const int64_t top = std::max<int64_t>(sp.dst_off[q] + h.send_ptr[q + 1] - h.send_ptr[q],
                                            rp.src_off[q] + h.recv_ptr[q + 1] - h.recv_ptr[q]);
      STORM_REQUIRE(top * 16 <= c->comm->seg_bytes,
                    "op_set_halo: %lld rows for rank %d exceed the peer window's segment of %lld bytes (raise window_bytes)",
                    (long long)top, h.nbr_rank[q], (long long)c->comm->seg_bytes);
    }
    // what each neighbour will send must be what this rank's plan receives from it: the counts travel through the
    // windows' all-reduce slots (one value per (sender, receiver) pair, summed: every pair is written by one rank only)
    // -- left to the first exchange's bounded polls when the ranks are more than the slots hold
    return STORM_HIP_OK;
  }
  std::vector<double> mine((size_t)h.n_nbrs), theirs((size_t)h.n_nbrs, -1.0);
  for (int q = 0; q < h.n_nbrs; ++q) mine[(size_t)q] = (double)(h.send_ptr[q + 1] - h.send_ptr[q]);
  if (c->comm->host_exchange) {
    std::vector<int64_t> one((size_t)h.n_nbrs + 1);
    for (int q = 0; q <= h.n_nbrs; ++q) one[(size_t)q] = q;
    const int rc = c->comm->host_exchange(c->comm->host_user, h.n_nbrs, h.nbr_rank.data(), one.data(), mine.data(),
                                          one.data(), theirs.data());
    if (rc != 0) STORM_FAIL(STORM_HIP_E_COMM, "host halo-exchange callback returned %d", rc);
  } else {
    double *d_buf = nullptr;
    HIP_TRY(hipMalloc((void **)&d_buf, sizeof(double) * 2 * (size_t)h.n_nbrs));
    HIP_TRY(hipMemcpyAsync(d_buf, mine.data(), sizeof(double) * (size_t)h.n_nbrs, hipMemcpyHostToDevice, c->comm_stream));
    ncclResult_t r = ncclGroupStart();
    for (int q = 0; q < h.n_nbrs && r == ncclSuccess; ++q) {
      r = ncclSend(d_buf + q, 1, ncclDouble, h.nbr_rank[q], c->comm->halo, c->comm_stream);
      if (r == ncclSuccess) r = ncclRecv(d_buf + h.n_nbrs + q, 1, ncclDouble, h.nbr_rank[q], c->comm->halo, c->comm_stream);
    }
    if (r == ncclSuccess) r = ncclGroupEnd();
    hipError_t e = hipMemcpyAsync(theirs.data(), d_buf + h.n_nbrs, sizeof(double) * (size_t)h.n_nbrs,
                                  hipMemcpyDeviceToHost, c->comm_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->comm_stream);
    (void)hipFree(d_buf);
    if (r != ncclSuccess) STORM_FAIL(STORM_HIP_E_COMM, "halo plan cross-check: %s", ncclGetErrorString(r));
    HIP_TRY(e);
  }
  for (int q = 0; q < h.n_nbrs; ++q) {
    const int64_t expect = h.recv_ptr[q + 1] - h.recv_ptr[q];
    STORM_REQUIRE((int64_t)theirs[(size_t)q] == expect,
                  "op_set_halo: rank %d will send %lld rows to rank %d, whose plan receives %lld from it",
                  h.nbr_rank[q], (long long)theirs[(size_t)q], c->rank, (long long)expect);
  }
  return STORM_HIP_OK;
}

int comm_check_error(storm_hip_ctx *c) { return ipc_check_error(c); }

// For kernels that reduce AND exchange in one launch: the device view of the windows and the next all-reduce epoch.
bool comm_ipc_next(storm_hip_ctx *c, IpcDev *w) {
  if (c->comm == nullptr || !c->comm->ipc) return false;
  *w = ipc_dev(c);
  return true;
}

void comm_destroy(storm_hip_ctx *c) {
  if (!c->comm) return;
  if (c->comm->red && c->comm->red != c->comm->halo) (void)ncclCommDestroy(c->comm->red);
  if (c->comm->halo) (void)ncclCommDestroy(c->comm->halo);
  if (c->comm->h_stage) (void)hipHostFree(c->comm->h_stage);
  if (c->comm->d_prof) (void)hipFree(c->comm->d_prof);
  if (c->comm->d_flag) {
    (void)hipFree(c->comm->d_flag);
    if (!c->comm->ipc && c->comm->h_error) (void)hipHostFree(c->comm->h_error);
  }
  if (c->comm->ipc || c->comm->win_local) {
    (void)hipDeviceSynchronize();
    for (int q = 0; q < (int)c->comm->win_peer.size(); ++q)
      if (c->comm->win_peer[(size_t)q] && c->comm->win_peer[(size_t)q] != c->comm->win_local)
        (void)hipIpcCloseMemHandle(c->comm->win_peer[(size_t)q]);
    (void)hipFree(c->comm->d_win_peer);
    (void)hipFree(c->comm->win_local);
    if (c->comm->h_error) (void)hipHostFree(c->comm->h_error);
    if (c->comm->d_stat) (void)hipFree(c->comm->d_stat);
  }
  delete c->comm;
  c->comm = nullptr;
}

}  // namespace storm

using namespace storm;

extern "C" {

int storm_hip_comm_unique_id(void *id128) {
  STORM_REQUIRE(id128, "comm_unique_id: null buffer");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return STORM_HIP_OK;
}

int storm_hip_ctx_comm_init(storm_hip_ctx *c, const void *id128, int n_ranks, int rank) {
  STORM_REQUIRE(c, "comm_init: null context");
  STORM_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "comm_init: rank %d of %d", rank, n_ranks);
  STORM_REQUIRE(c->comm == nullptr, "comm_init: communicator already initialised");
  c->n_ranks = n_ranks;
  c->rank = rank;
  if (n_ranks == 1 && id128 == nullptr) return STORM_HIP_OK;
  STORM_REQUIRE(id128, "comm_init: null unique id with %d ranks", n_ranks);
  HIP_TRY(hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  auto *cm = new Comm();
  ncclResult_t r = ncclCommInitRank(&cm->halo, n_ranks, id, rank);
  if (r != ncclSuccess) {
    delete cm;
    c->n_ranks = 1, c->rank = 0;
    STORM_FAIL(STORM_HIP_E_COMM, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
  }
  // Second communicator for the reductions; fall back to sharing one if the split is refused.  The decision
  // must be the SAME on every rank (a rank that all-reduces on `halo` while the others use `red` hangs the first
  // dot product), so the ranks agree on it: min over ranks of "my split worked", on the communicator all have.
  r = ncclCommSplit(cm->halo, 0, rank, &cm->red, nullptr);
  int split_ok = (r == ncclSuccess && cm->red != nullptr) ? 1 : 0;
  {
    int *d_flag = reinterpret_cast<int *>(c->d_scalars);
    bool agreed = hipMemcpyAsync(d_flag, &split_ok, sizeof(int), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
                  ncclAllReduce(d_flag, d_flag, 1, ncclInt, ncclMin, cm->halo, c->stream) == ncclSuccess &&
                  hipMemcpyAsync(&split_ok, d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                  hipStreamSynchronize(c->stream) == hipSuccess;
    if (!agreed) {
      if (cm->red && cm->red != cm->halo) (void)ncclCommDestroy(cm->red);
      (void)ncclCommDestroy(cm->halo);
      delete cm;
      c->n_ranks = 1, c->rank = 0;
      STORM_FAIL(STORM_HIP_E_COMM, "comm_init: the ranks could not agree on the reduction communicator");
    }
  }
  if (!split_ok) {
    if (cm->red && cm->red != cm->halo) (void)ncclCommDestroy(cm->red);
    cm->red = cm->halo;
  }
  // the flag of option rccl_flag_wait (and where a wait that timed out is reported); without them: cross-stream events
  if (hipMalloc((void **)&cm->d_flag, 256) == hipSuccess && hipMemset(cm->d_flag, 0, 256) == hipSuccess &&
      hipHostMalloc((void **)&cm->h_error, sizeof(int), hipHostMallocMapped) == hipSuccess) {
    *cm->h_error = 0;
    if (hipHostGetDevicePointer((void **)&cm->d_error, cm->h_error, 0) != hipSuccess) cm->d_error = nullptr;
  } else {
    (void)hipGetLastError();
    if (cm->d_flag) (void)hipFree(cm->d_flag);
    cm->d_flag = nullptr;
  }
  c->comm = cm;
  return STORM_HIP_OK;
}

int storm_hip_ctx_comm_init_host(storm_hip_ctx *c, int n_ranks, int rank, storm_hip_allreduce_fn allreduce,
                                 storm_hip_exchange_fn exchange, void *user) {
  STORM_REQUIRE(c, "comm_init_host: null context");
  STORM_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "comm_init_host: rank %d of %d", rank, n_ranks);
  STORM_REQUIRE(c->comm == nullptr, "comm_init_host: communicator already initialised");
  STORM_REQUIRE(allreduce && exchange, "comm_init_host: null callback");
  auto *cm = new Comm();
  cm->host_allreduce = allreduce, cm->host_exchange = exchange, cm->host_user = user;
  c->n_ranks = n_ranks, c->rank = rank;
  c->comm = cm;
  return STORM_HIP_OK;
}

int storm_hip_ctx_comm_ipc_export(storm_hip_ctx *c, int n_ranks, int rank, int64_t window_bytes, void *handle64) {
  STORM_REQUIRE(c && handle64, "comm_ipc_export: null argument");
  STORM_REQUIRE(n_ranks >= 1 && n_ranks <= 64 && rank >= 0 && rank < n_ranks, "comm_ipc_export: rank %d of %d (<= 64 ranks)", rank,
                n_ranks);
  STORM_REQUIRE(c->comm == nullptr, "comm_ipc_export: communicator already initialised");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
  HIP_TRY(hipSetDevice(c->device));
  // default: room for a 256 x 256 plane (16 bytes per value) from every rank, twice (parity)
  if (window_bytes <= 0) window_bytes = ipc_header_bytes(n_ranks) + (int64_t)2 * n_ranks * ((int64_t)17 << 16);
  auto *cm = new Comm();
  const int64_t P = n_ranks;
  const int64_t header = ipc_header_bytes(P);
  cm->seg_bytes = ((window_bytes - header) / (2 * P)) / 256 * 256;
  if (cm->seg_bytes < 256) {
    delete cm;
    STORM_FAIL(STORM_HIP_E_INVALID, "comm_ipc_export: a window of %lld bytes is too small for %d ranks", (long long)window_bytes,
               n_ranks);
  }
  cm->win_bytes = header + 2 * P * cm->seg_bytes;
  // Fine-grained device memory (what collective libraries allocate for their low-latency protocols): another GPU's
  // stores into it and this GPU's polling loads meet in memory while kernels run on both sides.  (An allocation the
  // runtime refuses that way falls back to hipMalloc: every access to the window is a system-scope `sc0 sc1` one.)
  hipError_t e = hipExtMallocWithFlags((void **)&cm->win_local, (size_t)cm->win_bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc((void **)&cm->win_local, (size_t)cm->win_bytes);
  }
  if (e == hipSuccess) e = hipMemset(cm->win_local, 0, (size_t)cm->win_bytes);
  hipIpcMemHandle_t handle;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&handle, cm->win_local);
  if (e != hipSuccess) {
    (void)hipFree(cm->win_local);
    delete cm;
    STORM_FAIL(STORM_HIP_E_COMM, "comm_ipc_export: %s", hipGetErrorString(e));
  }
  memcpy(handle64, &handle, sizeof handle);
  c->comm = cm;  // not usable before storm_hip_ctx_comm_init_ipc
  c->n_ranks = n_ranks, c->rank = rank;
  return STORM_HIP_OK;
}

int storm_hip_ctx_comm_init_ipc(storm_hip_ctx *c, const void *handles) {
  STORM_REQUIRE(c && handles, "comm_init_ipc: null argument");
  STORM_REQUIRE(c->comm != nullptr && c->comm->win_local != nullptr && !c->comm->ipc,
                "comm_init_ipc: call storm_hip_ctx_comm_ipc_export first");
  Comm *cm = c->comm;
  HIP_TRY(hipSetDevice(c->device));
  cm->win_peer.assign((size_t)c->n_ranks, nullptr);
  for (int q = 0; q < c->n_ranks; ++q) {
    if (q == c->rank) {
      cm->win_peer[(size_t)q] = cm->win_local;
      continue;
    }
    hipIpcMemHandle_t handle;
    memcpy(&handle, static_cast<const char *>(handles) + (size_t)q * sizeof handle, sizeof handle);
    void *mapped = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&mapped, handle, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess)
      STORM_FAIL(STORM_HIP_E_COMM, "comm_init_ipc: cannot map the window of rank %d: %s", q, hipGetErrorString(e));
    cm->win_peer[(size_t)q] = static_cast<char *>(mapped);
  }
  HIP_TRY(hipMalloc((void **)&cm->d_win_peer, sizeof(char *) * (size_t)c->n_ranks));
  HIP_TRY(hipMemcpy(cm->d_win_peer, cm->win_peer.data(), sizeof(char *) * (size_t)c->n_ranks, hipMemcpyHostToDevice));
  HIP_TRY(hipHostMalloc((void **)&cm->h_error, sizeof(int), hipHostMallocMapped));
  *cm->h_error = 0;
  HIP_TRY(hipHostGetDevicePointer((void **)&cm->d_error, cm->h_error, 0));
  HIP_TRY(hipMalloc((void **)&cm->d_stat, sizeof(long long) * 8));
  HIP_TRY(hipMemset(cm->d_stat, 0, sizeof(long long) * 8));
  cm->pair_epoch.assign((size_t)c->n_ranks, 0ull);
  cm->ipc = true;
  return STORM_HIP_OK;
}

int storm_hip_ctx_comm_size(storm_hip_ctx *c, int *n_ranks, int *rank) {
  STORM_REQUIRE(c, "comm_size: null context");
  if (n_ranks) *n_ranks = c->n_ranks;
  if (rank) *rank = c->rank;
  return STORM_HIP_OK;
}

int storm_hip_ctx_comm_rccl_view(storm_hip_ctx *c, int *halo_count, int *halo_user_rank, int *halo_device, int *red_count,
                                 int *hip_device, char *pci_bus_id, int pci_bus_id_len) {
  STORM_REQUIRE(c, "comm_rccl_view: null context");
  if (halo_count) *halo_count = 0;
  if (halo_user_rank) *halo_user_rank = -1;
  if (halo_device) *halo_device = -1;
  if (red_count) *red_count = 0;
  if (hip_device) *hip_device = c->device;
  if (pci_bus_id && pci_bus_id_len > 0) {
    pci_bus_id[0] = 0;
    if (hipDeviceGetPCIBusId(pci_bus_id, pci_bus_id_len, c->device) != hipSuccess) (void)hipGetLastError(), pci_bus_id[0] = 0;
  }
  if (!comm_is_rccl(c)) return STORM_HIP_OK;  // no RCCL communicator on this context: the counts stay 0
  int v = 0;
  if (halo_count) {
    NCCL_TRY(ncclCommCount(c->comm->halo, &v));
    *halo_count = v;
  }
  if (halo_user_rank) {
    NCCL_TRY(ncclCommUserRank(c->comm->halo, &v));
    *halo_user_rank = v;
  }
  if (halo_device) {
    NCCL_TRY(ncclCommCuDevice(c->comm->halo, &v));
    *halo_device = v;
  }
  if (red_count && c->comm->red) {
    NCCL_TRY(ncclCommCount(c->comm->red, &v));
    *red_count = v;
  }
  ncclResult_t async = ncclSuccess;
  NCCL_TRY(ncclCommGetAsyncError(c->comm->halo, &async));
  if (async != ncclSuccess) STORM_FAIL(STORM_HIP_E_COMM, "RCCL reports an asynchronous error on the halo communicator: %s", ncclGetErrorString(async));
  return STORM_HIP_OK;
}

int storm_hip_op_set_halo(storm_hip_op *op, int n_nbrs, const int32_t *nbr_rank, const int64_t *send_ptr,
                          const int64_t *send_idx, const int64_t *recv_ptr) {
  STORM_REQUIRE(op, "op_set_halo: null operator");
  STORM_REQUIRE(n_nbrs >= 0, "op_set_halo: negative neighbour count");
  STORM_REQUIRE(op->halo.n_nbrs == 0 && op->halo.d_send_idx == nullptr, "op_set_halo: plan already set");
  if (n_nbrs == 0) return STORM_HIP_OK;
  STORM_REQUIRE(nbr_rank && send_ptr && send_idx && recv_ptr, "op_set_halo: null array");
  storm_hip_ctx *c = op->ctx;
  HaloPlan &h = op->halo;
  STORM_REQUIRE(send_ptr[0] == 0 && recv_ptr[0] == 0, "op_set_halo: offsets must start at 0");
  for (int q = 0; q < n_nbrs; ++q) {
    STORM_REQUIRE(nbr_rank[q] >= 0 && nbr_rank[q] < c->n_ranks,  // a self-neighbour is a periodic boundary
                  "op_set_halo: neighbour %d is rank %d (this is rank %d of %d)", q, nbr_rank[q], c->rank, c->n_ranks);
    STORM_REQUIRE(send_ptr[q + 1] >= send_ptr[q] && recv_ptr[q + 1] >= recv_ptr[q], "op_set_halo: offsets not monotone");
  }
  STORM_REQUIRE(recv_ptr[n_nbrs] == op->n_halo, "op_set_halo: plan receives %lld rows, operator has %lld halo rows",
                (long long)recv_ptr[n_nbrs], (long long)op->n_halo);
  const int64_t n_send = send_ptr[n_nbrs];
  std::vector<int> idx((size_t)n_send);
  for (int64_t i = 0; i < n_send; ++i) {
    STORM_REQUIRE(send_idx[i] >= 0 && send_idx[i] < op->n_rows, "op_set_halo: send index %lld outside the owned rows",
                  (long long)send_idx[i]);
    idx[(size_t)i] = (int)send_idx[i];
  }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMalloc((void **)&h.d_send_idx, sizeof(int) * (size_t)(n_send ? n_send : 1)));
  HIP_TRY(hipMalloc((void **)&h.d_sendbuf, sizeof(double) * (size_t)(n_send ? n_send : 1)));
  if (n_send) HIP_TRY(hipMemcpy(h.d_send_idx, idx.data(), sizeof(int) * (size_t)n_send, hipMemcpyHostToDevice));
  h.n_nbrs = n_nbrs;
  h.n_send = n_send;
  h.nbr_rank.assign(nbr_rank, nbr_rank + n_nbrs);
  h.send_ptr.assign(send_ptr, send_ptr + n_nbrs + 1);
  h.recv_ptr.assign(recv_ptr, recv_ptr + n_nbrs + 1);
  STORM_TRY(halo_plan_cross_check(op));
  return op_upload_slice_lists(op);
}

}  // extern "C"
