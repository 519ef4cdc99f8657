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
  unsigned long long ar_epoch = 0, halo_epoch = 0;
  int *d_error = nullptr, *h_error = nullptr;  // set by a kernel whose wait timed out
  double *pending_x = nullptr;          // single-stream mode: the receive half runs in comm_halo_exchange_end
  unsigned long long pending_epoch = 0;
};

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

// ---- peer-window transport ---------------------------------------------------------------------------------
// Window of rank r (all offsets multiples of 256 bytes; P = n_ranks, "parity" = epoch & 1 double-buffers everything):
//   [all-reduce slots ]  2 x P x kIpcArSlot      slot (parity, s): the 64 values rank s contributed, each as two tagged 8-byte words
//   [halo flags       ]  2 x P x 64              flag (parity, s): epoch of the plane rank s has finished writing
//   [halo acks        ]  P x 64                  ack (d): last epoch rank d has consumed of what THIS rank sent it
//   [halo data        ]  2 x P x seg_bytes       data (parity, s): the rows rank s sends here
// One-shot all-reduce of <= 64 doubles: every rank writes its values, then (system-scope release) its tag, into slot
// (parity, rank) of EVERY window, polls its own window until all P tags carry the epoch and adds the values in rank
// order -- the same bits on every rank, two traversals of the link instead of RCCL's latency-bound ring / tree.
// The double buffer is safe without further handshakes: a rank can only start epoch e + 2 after finishing e + 1,
// which needed every peer's e + 1 contribution, which a peer sends after it has read epoch e.
// Halo: the sender's pack kernel stores x[send_idx] straight into data (parity, rank) of the RECEIVER's window once
// the receiver has acknowledged the plane that used this buffer two exchanges ago; a flag kernel publishes the
// epoch; the receiver's kernel polls the flag, copies the plane behind its owned rows and acknowledges.
// Every wait is bounded (kIpcTimeoutTicks of the 100 MHz real-time counter): a kernel that gives up sets
// *error and the host reports it at the next synchronisation instead of hanging.
__global__ __launch_bounds__(kBlock) void ipc_allreduce_kernel(IpcDev w, double *buf, int count,
                                                               unsigned long long epoch) {
  ipc_allreduce_block(w, buf, count, epoch);
}

// x[idx[i]], i < n  ->  the receiver's window, once it has consumed what this buffer held two exchanges ago.
// Plain 16-byte stores: the flag that publishes them is stored by a LATER kernel (a kernel's stores are complete and
// visible when it ends), so no per-store coherence is needed.
__global__ __launch_bounds__(kBlock) void ipc_halo_send_kernel(IpcDev w, int peer, int64_t n, int64_t dst_off,
                                                               const int *__restrict__ idx,
                                                               const double *__restrict__ x,
                                                               unsigned long long epoch) {
  if (threadIdx.x == 0 && epoch > 2)
    (void)ipc_wait_ge(reinterpret_cast<const unsigned long long *>(w.local + w.ack_off + (int64_t)peer * 64), epoch - 2,
                      w.error);
  __syncthreads();
  double *dst = reinterpret_cast<double *>(w.peers[peer] + w.data_off + ((int64_t)(epoch & 1) * w.n_ranks + w.rank) * w.seg_bytes) +
                dst_off;  // dst_off is even (segments and entry offsets are 16-byte aligned)
  typedef double double2v __attribute__((ext_vector_type(2)));
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    double2v v;
    v.x = x[idx[2 * i]], v.y = x[idx[2 * i + 1]];
    __builtin_nontemporal_store(v, reinterpret_cast<double2v *>(dst) + i);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = x[idx[n - 1]];
}
struct IpcPeers {
  int n;
  int rank[16];
};
__global__ void ipc_halo_flag_kernel(IpcDev w, IpcPeers peers, unsigned long long epoch) {
  // (the send kernels have completed: their write-through stores are acknowledged)
  if ((int)threadIdx.x < peers.n)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(w.peers[peers.rank[threadIdx.x]] + w.flag_off +
                                                              ((int64_t)(epoch & 1) * w.n_ranks + w.rank) * 64),
                       epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(kBlock) void ipc_halo_recv_kernel(IpcDev w, int peer, int64_t n, int64_t src_off,
                                                               double *__restrict__ x_halo,
                                                               unsigned long long epoch) {
  __shared__ int ok;
  if (threadIdx.x == 0)
    ok = ipc_wait_ge(reinterpret_cast<const unsigned long long *>(w.local + w.flag_off +
                                                                  ((int64_t)(epoch & 1) * w.n_ranks + peer) * 64),
                     epoch, w.error);
  __syncthreads();
  if (!ok) return;
  const double *src = reinterpret_cast<const double *>(w.local + w.data_off + ((int64_t)(epoch & 1) * w.n_ranks + peer) * w.seg_bytes) +
                      src_off;
  // system-coherent loads (the lines of this buffer that this XCD's L2 may still hold are two exchanges old)
  const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src + 2 * i) : "memory");
    x_halo[2 * i] = __hiloint2double((int)v.y, (int)v.x);
    x_halo[2 * i + 1] = __hiloint2double((int)v.w, (int)v.z);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x_halo[n - 1] = sys_load(src + n - 1);
}
__global__ void ipc_halo_ack_kernel(IpcDev w, IpcPeers peers, unsigned long long epoch) {
  if ((int)threadIdx.x < peers.n)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(w.peers[peers.rank[threadIdx.x]] + w.ack_off + (int64_t)w.rank * 64),
                       epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static IpcDev ipc_dev(const storm_hip_ctx *c) {
  const Comm *cm = c->comm;
  const int64_t P = c->n_ranks;
  IpcDev w;
  w.peers = cm->d_win_peer, w.local = cm->win_local, w.n_ranks = c->n_ranks, w.rank = c->rank;
  w.ar_off = 0;
  w.flag_off = 2 * P * kIpcArSlot;
  w.ack_off = w.flag_off + 2 * P * 64;
  w.data_off = (w.ack_off + P * 64 + 255) / 256 * 256;
  w.seg_bytes = cm->seg_bytes;
  w.error = cm->d_error;
  return w;
}
static int ipc_check_error(storm_hip_ctx *c) {
  if (c->comm && c->comm->ipc && *(volatile int *)c->comm->h_error != 0)
    STORM_FAIL(STORM_HIP_E_COMM, "peer-window transport: a wait for another rank timed out (rank %d of %d)", c->rank,
               c->n_ranks);
  return STORM_HIP_OK;
}
// distinct peer ranks of a plan and, per entry, where its rows start inside the (sender -> receiver) segment
static void ipc_plan_offsets(const HaloPlan &h, IpcPeers *peers, std::vector<int64_t> *send_off,
                             std::vector<int64_t> *recv_off) {
  peers->n = 0;
  send_off->assign((size_t)h.n_nbrs, 0), recv_off->assign((size_t)h.n_nbrs, 0);
  for (int q = 0; q < h.n_nbrs; ++q) {
    bool seen = false;
    for (int q2 = 0; q2 < q; ++q2)
      if (h.nbr_rank[q2] == h.nbr_rank[q]) {
        seen = true;
        (*send_off)[(size_t)q] += (h.send_ptr[q2 + 1] - h.send_ptr[q2] + 1) & ~(int64_t)1;  // 16-byte aligned entries
        (*recv_off)[(size_t)q] += (h.recv_ptr[q2 + 1] - h.recv_ptr[q2] + 1) & ~(int64_t)1;
      }
    if (!seen && peers->n < 16) peers->rank[peers->n++] = h.nbr_rank[q];
  }
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
    hipLaunchKernelGGL(ipc_allreduce_kernel, dim3(1), dim3(kBlock), 0, c->stream, ipc_dev(c), d_buf, count,
                       ++c->comm->ar_epoch);
    HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
  }
  STORM_REQUIRE(c->comm && c->comm->red, "all-reduce without an initialised communicator");
  NCCL_TRY(ncclAllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum, c->comm->red, c->stream));
  return STORM_HIP_OK;
}

__global__ __launch_bounds__(kBlock) void halo_pack_kernel(int64_t n, const int *__restrict__ idx,
                                                           const double *__restrict__ x,
                                                           double *__restrict__ buf) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) buf[i] = x[idx[i]];
}

static int ipc_halo_receive(const storm_hip_op *op, double *x, unsigned long long epoch, hipStream_t hs) {
  storm_hip_ctx *c = op->ctx;
  const HaloPlan &h = op->halo;
  const IpcDev w = ipc_dev(c);
  IpcPeers peers;
  std::vector<int64_t> send_off, recv_off;
  ipc_plan_offsets(h, &peers, &send_off, &recv_off);
  for (int q = 0; q < h.n_nbrs; ++q) {
    const int64_t nr = h.recv_ptr[q + 1] - h.recv_ptr[q];
    if (nr <= 0) continue;
    const int nb = (int)std::min<int64_t>(512, (nr + kBlock * 2 - 1) / (kBlock * 2));
    hipLaunchKernelGGL(ipc_halo_recv_kernel, dim3(nb), dim3(kBlock), 0, hs, w, h.nbr_rank[q], nr, recv_off[(size_t)q],
                       x + op->n_rows + h.recv_ptr[q], epoch);
  }
  hipLaunchKernelGGL(ipc_halo_ack_kernel, dim3(1), dim3(kWave), 0, hs, w, peers, epoch);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
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
    Comm *cm = c->comm;
    STORM_TRY(ipc_check_error(c));
    const IpcDev w = ipc_dev(c);
    const unsigned long long epoch = ++cm->halo_epoch;
    IpcPeers peers;
    std::vector<int64_t> send_off, recv_off;
    ipc_plan_offsets(h, &peers, &send_off, &recv_off);
    // Two streams (default): the exchange runs on the comm stream beside the interior rows, ordered by events.  One
    // stream (option ipc_streams = 1): send + flag ahead of the interior launch, receive + acknowledge behind it on
    // the compute stream -- no cross-stream events; the sends are then not hidden behind the interior rows.
    const bool one_stream = c->opt_ipc_streams == 1;
    hipStream_t hs = one_stream ? c->stream : c->comm_stream;
    if (!one_stream) {
      HIP_TRY(hipEventRecord(c->ev_x_ready, c->stream));  // x must be complete before it is packed
      HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_x_ready, 0));
    }
    for (int q = 0; q < h.n_nbrs; ++q) {
      const int64_t ns = h.send_ptr[q + 1] - h.send_ptr[q];
      if (ns <= 0) continue;
      const int nb = (int)std::min<int64_t>(512, (ns + kBlock * 2 - 1) / (kBlock * 2));
      hipLaunchKernelGGL(ipc_halo_send_kernel, dim3(nb), dim3(kBlock), 0, hs, w, h.nbr_rank[q], ns,
                         send_off[(size_t)q], h.d_send_idx + h.send_ptr[q], (const double *)x, epoch);
    }
    hipLaunchKernelGGL(ipc_halo_flag_kernel, dim3(1), dim3(kWave), 0, hs, w, peers, epoch);
    HIP_TRY(hipGetLastError());
    if (one_stream) {
      cm->pending_x = x, cm->pending_epoch = epoch;
      return STORM_HIP_OK;
    }
    STORM_TRY(ipc_halo_receive(op, x, epoch, hs));
    HIP_TRY(hipEventRecord(c->ev_halo_done, c->comm_stream));
    return STORM_HIP_OK;
  }
  STORM_REQUIRE(c->comm && c->comm->halo, "halo exchange without an initialised communicator");
  // x must be complete before it is packed
  HIP_TRY(hipEventRecord(c->ev_x_ready, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_x_ready, 0));
  if (h.n_send > 0) {
    const int64_t need = (h.n_send + kBlock - 1) / kBlock;
    const int nb = (int)(need > 1024 ? 1024 : need);
    hipLaunchKernelGGL(halo_pack_kernel, dim3(nb), dim3(kBlock), 0, c->comm_stream, h.n_send, h.d_send_idx, x,
                       h.d_sendbuf);
    HIP_TRY(hipGetLastError());
  }
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
  HIP_TRY(hipEventRecord(c->ev_halo_done, c->comm_stream));
  return STORM_HIP_OK;
}

int comm_halo_exchange_end(const storm_hip_op *op) {
  storm_hip_ctx *c = op->ctx;
  if (op->halo.n_nbrs == 0 || c->comm == nullptr || c->comm->host_exchange) return STORM_HIP_OK;
  if (c->comm->ipc && c->comm->pending_x != nullptr) {  // single-stream mode: the receive half, behind the interior rows
    double *x = c->comm->pending_x;
    c->comm->pending_x = nullptr;
    return ipc_halo_receive(op, x, c->comm->pending_epoch, c->stream);
  }
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_halo_done, 0));
  return STORM_HIP_OK;
}


// A plan whose send count towards a neighbour differs from what that neighbour expects to receive hangs the first
// exchange inside RCCL.  Once, when the plan is set: every rank tells each neighbour how many rows it will send,
// and compares what it is told with its own receive counts.  (Same transport as the halo itself.)
int halo_plan_cross_check(const storm_hip_op *op) {
  storm_hip_ctx *c = op->ctx;
  const HaloPlan &h = op->halo;
  if (h.n_nbrs == 0 || c->comm == nullptr) return STORM_HIP_OK;
  if (c->comm->ipc) {  // every (sender -> receiver) pair owns one segment of the receiver's window
    IpcPeers peers;
    std::vector<int64_t> send_off, recv_off;
    ipc_plan_offsets(h, &peers, &send_off, &recv_off);
    int distinct = 0;
    for (int q = 0; q < h.n_nbrs; ++q) {
      bool seen = false;
      for (int q2 = 0; q2 < q; ++q2) seen |= h.nbr_rank[q2] == h.nbr_rank[q];
      distinct += !seen;
      const int64_t top = std::max(send_off[(size_t)q] + h.send_ptr[q + 1] - h.send_ptr[q],
                                   recv_off[(size_t)q] + h.recv_ptr[q + 1] - h.recv_ptr[q]);
      STORM_REQUIRE(top * 8 <= c->comm->seg_bytes,
                    "op_set_halo: %lld rows for rank %d exceed the peer window's segment of %lld bytes (raise window_bytes)",
                    (long long)top, h.nbr_rank[q], (long long)c->comm->seg_bytes);
    }
    STORM_REQUIRE(distinct <= 16, "op_set_halo: %d neighbour ranks (the peer-window transport handles 16)", distinct);
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
bool comm_ipc_next(storm_hip_ctx *c, IpcDev *w, unsigned long long *epoch) {
  if (c->comm == nullptr || !c->comm->ipc) return false;
  *w = ipc_dev(c);
  *epoch = ++c->comm->ar_epoch;
  return true;
}

void comm_destroy(storm_hip_ctx *c) {
  if (!c->comm) return;
  if (c->comm->red && c->comm->red != c->comm->halo) (void)ncclCommDestroy(c->comm->red);
  if (c->comm->halo) (void)ncclCommDestroy(c->comm->halo);
  if (c->comm->h_stage) (void)hipHostFree(c->comm->h_stage);
  if (c->comm->ipc || c->comm->win_local) {
    (void)hipDeviceSynchronize();
    for (int q = 0; q < (int)c->comm->win_peer.size(); ++q)
      if (c->comm->win_peer[(size_t)q] && c->comm->win_peer[(size_t)q] != c->comm->win_local)
        (void)hipIpcCloseMemHandle(c->comm->win_peer[(size_t)q]);
    (void)hipFree(c->comm->d_win_peer);
    (void)hipFree(c->comm->win_local);
    if (c->comm->h_error) (void)hipHostFree(c->comm->h_error);
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
  if (window_bytes <= 0) window_bytes = (int64_t)16 << 20;
  auto *cm = new Comm();
  const int64_t P = n_ranks;
  const int64_t header = ((2 * P * kIpcArSlot + 2 * P * 64 + P * 64) + 255) / 256 * 256;
  cm->seg_bytes = ((window_bytes - header) / (2 * P)) / 256 * 256;
  if (cm->seg_bytes < 256) {
    delete cm;
    STORM_FAIL(STORM_HIP_E_INVALID, "comm_ipc_export: a window of %lld bytes is too small for %d ranks", (long long)window_bytes,
               n_ranks);
  }
  cm->win_bytes = header + 2 * P * cm->seg_bytes;
  hipError_t e = hipMalloc((void **)&cm->win_local, (size_t)cm->win_bytes);
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
  cm->ipc = true;
  return STORM_HIP_OK;
}

int storm_hip_ctx_comm_size(storm_hip_ctx *c, int *n_ranks, int *rank) {
  STORM_REQUIRE(c, "comm_size: null context");
  if (n_ranks) *n_ranks = c->n_ranks;
  if (rank) *rank = c->rank;
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
