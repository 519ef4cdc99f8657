// Held-back statements for HOST loops (option lazy_statements): an iterate() written statement by statement against the
// reference's interface -- `any_op.mul(z, p); alpha = safe_divide(gamma, dot_product(p, z)); x += alpha * p;
// r -= alpha * z; gamma = dot_product(r, r); p <<= r + beta * p;` (SolverCg.hpp:96-123) -- pays a pass over memory and a
// launch for every statement, and a second pass for every reduction.  With the option on, the vector statements of the
// overload census that are plain linear combinations (`<<=`, `+=`, `-=`, `*=`: storm_hip_copy / _scale / _axpy / _xpay /
// _axpbz) and storm_hip_op_apply are NOT launched when called: they wait, in program order, for the call that needs
// their result.  When that call is a reduction over a vector the LAST waiting statement writes, the reduction rides in
// that statement's kernel:
//     x += alpha p;  r -= alpha z;  <r, r>       ONE pass: both updates and the sum (lazy_lin_kernel<2, true>)
//     z = A p;  <p, z>                           the apply with its fused-dot epilogue (spmv_launch with SpmvDot)
//     x += alpha p;  [r -= alpha z;  <r, r>;]  p <<= r + beta p;  z = A p;  <p, z>
//                                                on a lattice operator the library's own fused CG step
//                                                (cg_step_march_kernel: x, p', z = A p' and the sum in one pass):
//                                                a statement the reduction does not depend on keeps waiting
//                                                (x += alpha p above), and p' is written to a spare vector whose
//                                                storage then BECOMES p's (never for a vector whose address
//                                                storm_hip_vec_device_ptr has handed out)
// Two consecutive linear statements always leave as one pass (statement 2 sees statement 1's values: elementwise, in
// order).  Every other entry point of the library launches the waiting statements first (lazy_sync), so nothing is ever
// observed out of order; with the option off the queue is always empty.  Arithmetic: for the linear statements and their
// reductions the expressions, the block / thread mapping and the summation trees of the eager kernels (blas1.hip
// ew_kernel<AxpbzF>, multi_dot_ticket_kernel) -- the same bits, statement by statement; the apply's fused dot sums in the
// SpMV kernel's own order (per wave, as in the library's fused solver loops), and the fused CG step rounds x + alpha p
// and r + beta p once (an FMA, as the library's device loop does) where the statements round the product first: equal to
// rounding (tests/test_gpu_lazy.py).
#include <cmath>
#include <cstring>

#include "common.hpp"
#include "blas1_device.hpp"
#include "ticket_device.hpp"

namespace storm {

struct LazyLin {
  double *y;
  const double *v[2];
  double c[2];
  int nt;      // terms: 1 or 2
  int src[2];  // -1: load v[t]; 0: the value statement 0 of this launch produced for the row
};
struct LazyArgs {
  LazyLin s[2];
  const double *da, *db;  // operands of the reduction <da, db>
  int sa, sb;             // -1: load; k: the value statement k produced
};

__device__ __forceinline__ double lazy_eval(const LazyLin &L, double x0, double x1) {
  return L.nt == 2 ? L.c[0] * x0 + L.c[1] * x1 : L.c[0] * x0;  // (AxpbzF's expression: blas1.hip)
}

// NS statements over the rows, in order, then (DOT) <da, db> finished by tickets: the sum to out[0] and, as two
// self-validating words, to pinned host memory.  One trip per thread, kUnroll row pairs per stream in flight.
template <int NS, bool DOT>
__global__ __launch_bounds__(kBlock) void lazy_lin_kernel(int64_t n, LazyArgs A, TicketArgs tickets, double *__restrict__ out, int nt,
                                                          unsigned long long *host_words, unsigned tag) {
  __shared__ double lds[1][4];
  const unsigned bx = blockIdx.x;
  const int64_t n2 = n >> 1;
  double acc[1] = {0.0};
  nt_dispatch(nt, [&](auto ntc) {
    for (int64_t base = (int64_t)bx * (kBlock * kUnroll) + threadIdx.x; base < n2; base += (int64_t)gridDim.x * (kBlock * kUnroll)) {
      double2v in[kUnroll][NS][2], val[kUnroll][NS], da[kUnroll], db[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = base + u * kBlock;
        if (i < n2) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
              if (t < A.s[s].nt && A.s[s].src[t] < 0) in[u][s][t] = ld2(reinterpret_cast<const double2v *>(A.s[s].v[t]) + i, ntc);
          if (DOT) {
            if (A.sa < 0) da[u] = ld2(reinterpret_cast<const double2v *>(A.da) + i, ntc);
            if (A.sb < 0 && A.db != A.da) db[u] = ld2(reinterpret_cast<const double2v *>(A.db) + i, ntc);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t i = base + u * kBlock;
        if (i < n2) {
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            const LazyLin &L = A.s[s];
            const double2v x0 = L.src[0] >= 0 ? val[u][0] : in[u][s][0];
            const double2v x1 = L.nt == 2 ? (L.src[1] >= 0 ? val[u][0] : in[u][s][1]) : x0;
            double2v o;
            o.x = lazy_eval(L, x0.x, x1.x), o.y = lazy_eval(L, x0.y, x1.y);
            val[u][s] = o;
            st2(reinterpret_cast<double2v *>(L.y) + i, o, ntc);
          }
          if (DOT) {
            const double2v a = A.sa >= 0 ? val[u][A.sa >= NS ? NS - 1 : A.sa] : da[u];
            const double2v b = A.sb >= 0 ? val[u][A.sb >= NS ? NS - 1 : A.sb] : (A.db == A.da ? a : db[u]);
            acc[0] += a.x * b.x;
            acc[0] += a.y * b.y;
          }
        }
      }
    }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {  // the odd last row
    const int64_t i = n - 1;
    double val[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const LazyLin &L = A.s[s];
      const double x0 = L.src[0] >= 0 ? val[0] : L.v[0][i];
      const double x1 = L.nt == 2 ? (L.src[1] >= 0 ? val[0] : L.v[1][i]) : x0;
      val[s] = lazy_eval(L, x0, x1);
      L.y[i] = val[s];
    }
    if (DOT) {
      const double a = A.sa >= 0 ? val[A.sa >= NS ? NS - 1 : A.sa] : A.da[i];
      const double b = A.sb >= 0 ? val[A.sb >= NS ? NS - 1 : A.sb] : A.db[i];
      acc[0] += a * b;
    }
  }
  if (!DOT) return;
  double mine[1], total[1];
  block_sum_multi<1>(acc, lds, mine);
  if (threadIdx.x >= kWave) return;
  if (ticket_reduce_wave0<1>(tickets, mine, 1, bx, gridDim.x, total) && threadIdx.x == 0) {
    out[0] = total[0];
    if (host_words) {
      const unsigned long long t = (unsigned long long)tag << 32;
      __hip_atomic_store(host_words, t | (unsigned)__double2loint(total[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(host_words + 1, t | (unsigned)__double2hiint(total[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void lazy_result_kernel(const double *value, unsigned long long *host_words, unsigned tag) {
  const double v = *value;
  const unsigned long long t = (unsigned long long)tag << 32;
  __hip_atomic_store(host_words, t | (unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(host_words + 1, t | (unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

namespace {

constexpr int kLazySlot = kResultRing - 1;  // the result slot this unit polls (multi_dot_begin never has all eight in use here: see lazy_try_dot)

// The src codes of statement s: a term that reads what an earlier statement OF THE SAME LAUNCH wrote takes the value from
// registers (the store has not happened for the row yet as far as this thread's loads are concerned).
void resolve(LazyArgs &A, int ns) {
  for (int s = 0; s < ns; ++s)
    for (int t = 0; t < 2; ++t) {
      A.s[s].src[t] = -1;
      for (int e = 0; e < s; ++e)
        if (t < A.s[s].nt && A.s[s].v[t] == A.s[e].y) A.s[s].src[t] = e;
    }
}

int launch_lins(storm_hip_ctx *c, const LazyStmt *q, int ns, const double *da, const double *db, double *result) {
  const int64_t n = q[0].n;
  LazyArgs A{};
  for (int s = 0; s < ns; ++s) {
    A.s[s].y = q[s].y, A.s[s].nt = q[s].nt;
    for (int t = 0; t < 2; ++t) A.s[s].v[t] = q[s].v[t], A.s[s].c[t] = q[s].c[t];
  }
  resolve(A, ns);
  const bool dot = da != nullptr;
  A.da = da, A.db = db, A.sa = A.sb = -1;
  if (dot)
    for (int s = 0; s < ns; ++s) {  // (the LAST writer of the operand)
      if (da == q[s].y) A.sa = s;
      if (db == q[s].y) A.sb = s;
    }
  if (n <= 0) {
    if (dot) *result = 0.0;
    return STORM_HIP_OK;
  }
  int nb = stream_blocks(n);
  if ((int64_t)nb > c->partials_capacity) nb = (int)c->partials_capacity;
  const int nt = stream_nt(c, n);
  const TicketArgs t{c->d_tickets, c->d_partials, c->d_ticket_sums};
  const dim3 g(nb), b(kBlock);
  unsigned tag = 0;
  unsigned long long *words = nullptr;
  if (dot) {
    tag = ++c->result_seq ? c->result_seq : ++c->result_seq;
    words = c->d_result_words + 16 * kLazySlot;
  }
  if (ns == 1) {
    if (dot) hipLaunchKernelGGL((lazy_lin_kernel<1, true>), g, b, 0, c->stream, n, A, t, c->d_scalars, nt, words, tag);
    else hipLaunchKernelGGL((lazy_lin_kernel<1, false>), g, b, 0, c->stream, n, A, t, c->d_scalars, nt, words, tag);
  } else {
    if (dot) hipLaunchKernelGGL((lazy_lin_kernel<2, true>), g, b, 0, c->stream, n, A, t, c->d_scalars, nt, words, tag);
    else hipLaunchKernelGGL((lazy_lin_kernel<2, false>), g, b, 0, c->stream, n, A, t, c->d_scalars, nt, words, tag);
  }
  HIP_TRY(hipGetLastError());
  if (ns == 2) ++c->n_lazy_fused_pairs;
  if (!dot) return STORM_HIP_OK;
  ++c->n_lazy_fused_dots;
  volatile unsigned long long *w = c->h_result_words + 16 * kLazySlot;
  for (long spin = 0; (unsigned)(w[0] >> 32) != tag || (unsigned)(w[1] >> 32) != tag; ++spin)
    if ((spin & 0x3fff) == 0x3fff && hipStreamQuery(c->stream) == hipSuccess &&
        ((unsigned)(w[0] >> 32) != tag || (unsigned)(w[1] >> 32) != tag)) {
      // the stream is idle and the words never came (a failed launch?): the sum is in device memory if the kernel ran
      HIP_TRY(hipMemcpy(result, c->d_scalars, sizeof(double), hipMemcpyDeviceToHost));
      return STORM_HIP_OK;
    }
  const unsigned long long bits = (w[1] << 32) | (w[0] & 0xffffffffull);
  memcpy(result, &bits, sizeof(double));
  return STORM_HIP_OK;
}

// Launch the linear statements q[0 .. count): in pairs.
int launch_plain(storm_hip_ctx *c, const LazyStmt *q, int count) {
  for (int i = 0; i < count;) {
    const int ns = (i + 1 < count && q[i + 1].n == q[i].n) ? 2 : 1;
    STORM_TRY(launch_lins(c, q + i, ns, nullptr, nullptr, nullptr));
    i += ns;
  }
  return STORM_HIP_OK;
}

int launch_apply(storm_hip_ctx *c, const LazyStmt &a) {
  return spmv_launch(a.op, host_scal(a.alpha), host_scal(a.beta), a.x, a.y, nullptr, nullptr);
}

// Two linear statements that may run in either order: neither writes what the other reads or writes.
bool commute(const LazyStmt &s, const LazyStmt &t) {
  if (s.y == t.y) return false;
  for (int k = 0; k < 2; ++k)
    if ((k < t.nt && t.v[k] == s.y) || (k < s.nt && s.v[k] == t.y)) return false;
  return true;
}

// the sum a kernel left in device memory, to the host as two self-validating words the host polls (no copy, no stream
// wait: ~10 us less per reduction)
int fetch_sum(storm_hip_ctx *c, const double *d_value, double *result) {
  const unsigned tag = ++c->result_seq ? c->result_seq : ++c->result_seq;
  hipLaunchKernelGGL(lazy_result_kernel, dim3(1), dim3(1), 0, c->stream, d_value, c->d_result_words + 16 * kLazySlot, tag);
  HIP_TRY(hipGetLastError());
  volatile unsigned long long *hw = c->h_result_words + 16 * kLazySlot;
  for (long spin = 0; (unsigned)(hw[0] >> 32) != tag || (unsigned)(hw[1] >> 32) != tag; ++spin)
    if ((spin & 0x3fff) == 0x3fff && hipStreamQuery(c->stream) == hipSuccess &&
        ((unsigned)(hw[0] >> 32) != tag || (unsigned)(hw[1] >> 32) != tag)) {
      HIP_TRY(hipMemcpy(result, d_value, sizeof(double), hipMemcpyDeviceToHost));
      return STORM_HIP_OK;
    }
  const unsigned long long bits = (hw[1] << 32) | (hw[0] & 0xffffffffull);
  memcpy(result, &bits, sizeof(double));
  return STORM_HIP_OK;
}

// q = [..., x += ca p, p = r + cb p, z = beta p + alpha M(p)] and the sum asked for is <p, z>, M a lattice operator the
// marching kernel takes: the library's fused CG step.  p' goes to the spare vector, whose storage p's handle takes over.
// false: not that shape (nothing launched, q untouched).
bool try_cg_step(storm_hip_ctx *c, std::vector<LazyStmt> &q, const double *a, const double *b, double *result, int *status) {
  if (q.size() < 3 || c->opt_fuse_dot == 0 || c->opt_cg_fuse == 0 || c->opt_lazy < 2) return false;  // (lazy_statements = 2)
  const LazyStmt &ap = q[q.size() - 1], &sp = q[q.size() - 2], &sx = q[q.size() - 3];
  if (ap.kind != 1 || sp.kind != 0 || sx.kind != 0) return false;
  const int64_t n = ap.n;
  double *p = sp.y, *x = sx.y, *z = ap.y;
  const double *r = sp.v[0];
  if (sp.n != n || sx.n != n || sp.nt != 2 || sx.nt != 2) return false;
  if (sp.v[1] != p || sp.c[0] != 1.0 || r == p) return false;        // p = r + cb p
  if (sx.v[1] != x || sx.c[1] != 1.0 || sx.v[0] != p) return false;  // x = ca p + x
  if (ap.x != p || !((a == p && b == z) || (a == z && b == p))) return false;
  if (x == p || x == r || x == z || z == p || z == r) return false;
  storm_hip_vec *pv = sp.yvec;
  if (pv == nullptr || pv->d != p || pv->exposed || pv->n_halo != 0 || pv->n_owned != n || !spmv_can_march(ap.op)) return false;
  if (c->lazy_spare != nullptr && (c->lazy_spare->n_owned != pv->n_owned || c->lazy_spare->bytes != pv->bytes)) {
    (void)storm_hip_vec_destroy(c->lazy_spare);
    c->lazy_spare = nullptr;
  }
  if (c->lazy_spare == nullptr && vec_create_work(pv, &c->lazy_spare) != STORM_HIP_OK) return false;
  const LazyStmt step[3] = {sx, sp, ap};
  c->lazy_q.assign(q.begin(), q.end() - 3);
  *status = lazy_flush(c);
  if (*status != STORM_HIP_OK) return true;
  int nblocks = 0, ticketed = 0;
  SpmvDot sd;
  sd.w = p, sd.yy = false, sd.partials = c->d_partials, sd.nblocks_out = &nblocks;
  sd.out[0] = c->d_scalars, sd.ticketed_out = &ticketed;
  sd.cg.x = x, sd.cg.r = r, sd.cg.p_out = c->lazy_spare->d, sd.cg.ca_imm = step[0].c[0], sd.cg.cb_imm = step[1].c[1];
  *status = spmv_launch(step[2].op, host_scal(step[2].alpha), host_scal(step[2].beta), p, z, &sd, nullptr);
  if (*status != STORM_HIP_OK) return true;
  std::swap(pv->d, c->lazy_spare->d), std::swap(pv->base, c->lazy_spare->base);  // (same size, same pool key)
  if (!ticketed) *status = k_reduce_final(c, c->d_partials, nblocks, 1, c->d_scalars, nullptr);
  if (*status == STORM_HIP_OK) *status = fetch_sum(c, c->d_scalars, result);
  ++c->n_lazy_cg_steps;
  return true;
}

}  // namespace

int lazy_flush(storm_hip_ctx *c) {
  if (c->lazy_q.empty()) return STORM_HIP_OK;
  std::vector<LazyStmt> q;
  q.swap(c->lazy_q);  // (nothing launched below may find the queue non-empty)
  size_t i = 0;
  while (i < q.size()) {
    if (q[i].kind == 1) {
      STORM_TRY(launch_apply(c, q[i]));
      ++i;
      continue;
    }
    size_t j = i;
    while (j < q.size() && q[j].kind == 0) ++j;
    STORM_TRY(launch_plain(c, q.data() + i, (int)(j - i)));
    i = j;
  }
  return STORM_HIP_OK;
}

int lazy_push_lin(storm_hip_ctx *c, storm_hip_vec *yv, double c0, const double *v0, double c1, const double *v1, int nt, int64_t n) {
  // at most two linear statements wait, and none behind an apply (the apply's x may be what this one writes)
  if (!c->lazy_q.empty() && (c->lazy_q.back().kind == 1 || c->lazy_q.size() >= 2 || c->lazy_q.back().n != n)) STORM_TRY(lazy_flush(c));
  double *y = yv->d;
  LazyStmt s;
  s.yvec = yv;
  s.kind = 0, s.y = y, s.v[0] = v0, s.v[1] = nt == 2 ? v1 : nullptr, s.c[0] = c0, s.c[1] = nt == 2 ? c1 : 0.0, s.nt = nt, s.n = n;
  c->lazy_q.push_back(s);
  return STORM_HIP_OK;
}

int lazy_push_apply(const storm_hip_op *op, double alpha, double beta, const double *x, double *y) {
  storm_hip_ctx *c = op->ctx;
  if (!c->lazy_q.empty() && c->lazy_q.back().kind == 1) STORM_TRY(lazy_flush(c));
  LazyStmt s;
  s.kind = 1, s.op = op, s.alpha = alpha, s.beta = beta, s.x = x, s.y = y, s.n = op->n_rows;
  c->lazy_q.push_back(s);
  return STORM_HIP_OK;
}

// <a, b> when statements wait.  true: handled here (*status, *result set); false: the queue has been launched (or was
// empty) and the caller computes the reduction the ordinary way.
bool lazy_try_dot(storm_hip_ctx *c, const double *a, const double *b, int64_t n, double *result, int *status) {
  *status = STORM_HIP_OK;
  if (c->lazy_q.empty()) return false;
  const bool direct = c->comm == nullptr && c->opt_host_result != 0 && c->opt_ticket_reduce != 0 && c->api_done == nullptr &&
                      c->result_ring[kLazySlot].tag == 0;
  const LazyStmt last = c->lazy_q.back();
  if (!direct || last.n != n || (a != last.y && b != last.y)) {
    *status = lazy_flush(c);
    return false;
  }
  std::vector<LazyStmt> q;
  q.swap(c->lazy_q);
  if (last.kind == 0) {
    // everything before the last one or two linear statements goes out first; those ride with the reduction
    size_t first = q.size() - 1;
    if (first > 0 && q[first - 1].kind == 0 && q[first - 1].n == n) --first;
    if (first + 2 == q.size() && first == 0 && commute(q[0], q[1]) && a != q[0].y && b != q[0].y) {
      // the statement in front has nothing to do with the last one or the sum: it keeps waiting (the same bytes
      // either way; what follows may take it along -- the fused CG step above)
      *status = launch_lins(c, q.data() + 1, 1, a, b, result);
      c->lazy_q.assign(q.begin(), q.begin() + 1);
      return true;
    }
    c->lazy_q.assign(q.begin(), q.begin() + (long)first);
    *status = lazy_flush(c);
    if (*status == STORM_HIP_OK) *status = launch_lins(c, q.data() + first, (int)(q.size() - first), a, b, result);
    return true;
  }
  if (try_cg_step(c, q, a, b, result, status)) return true;
  // an apply: everything before it goes out, then y = beta x + alpha M(x) with <w, y> (w the other operand, or y itself)
  c->lazy_q.assign(q.begin(), q.end() - 1);
  *status = lazy_flush(c);
  if (*status != STORM_HIP_OK) return true;
  const double *w = (a == last.y) ? b : a;
  const bool yy = (w == last.y);
  if (last.op->tail_rows != 0 || c->opt_fuse_dot == 0) {  // (no fused epilogue for operators with a CSR tail)
    *status = launch_apply(c, last);
    return false;
  }
  int nblocks = 0, ticketed = 0;
  SpmvDot sd;
  sd.w = yy ? last.x : w;  // (the kernels want a w: with <y, y> alone its partial is computed and dropped)
  sd.yy = yy, sd.partials = c->d_partials, sd.nblocks_out = &nblocks;
  sd.out[0] = c->d_scalars, sd.out[1] = c->d_scalars + 1, sd.ticketed_out = &ticketed;
  *status = spmv_launch(last.op, host_scal(last.alpha), host_scal(last.beta), last.x, last.y, &sd, nullptr);
  if (*status != STORM_HIP_OK) return true;
  if (nblocks <= 0) return false;  // the launch did not fuse after all: the ordinary reduction follows
  if (!ticketed) *status = k_reduce_final(c, c->d_partials, nblocks, yy ? 2 : 1, c->d_scalars, nullptr);
  if (*status != STORM_HIP_OK) return true;
  *status = fetch_sum(c, c->d_scalars + (yy ? 1 : 0), result);
  ++c->n_lazy_apply_dots;
  return true;
}

}  // namespace storm
