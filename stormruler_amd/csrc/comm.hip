// Multi-GPU plumbing: RCCL over xGMI.
//
// One process per GPU.  Two communicators per context so the two traffic classes never
// queue behind each other: `halo` carries the point-to-point halo planes on the comm stream
// (overlapped with the interior rows of the SpMV on the compute stream), `red` carries the
// 8..400-byte dot-product all-reduces on the compute stream (latency-bound, so reductions
// are batched by the solvers: BiCGStab's (<t,r>,<t,t>) and (|r|^2,<rt,r>) are one call each).
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "common.hpp"

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

void comm_destroy(storm_hip_ctx *c) {
  if (!c->comm) return;
  if (c->comm->red && c->comm->red != c->comm->halo) (void)ncclCommDestroy(c->comm->red);
  if (c->comm->halo) (void)ncclCommDestroy(c->comm->halo);
  if (c->comm->h_stage) (void)hipHostFree(c->comm->h_stage);
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
