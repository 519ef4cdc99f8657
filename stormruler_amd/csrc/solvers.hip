// Device-resident Krylov loops: CG, BiCGStab, GMRES(m).
//
// Restates Solvers/Solver.hpp:116-147 (IterativeSolver::solve), :236-257
// (InnerOuterIterativeSolver), SolverCg.hpp:54-126, SolverBiCgStab.hpp:59-165 and
// SolverGmres.hpp:51-249 for the operator A = beta*I + alpha*M on the device.
//
// Design: every scalar of the recurrences (gamma, alpha, beta, rho, omega, the Hessenberg
// column, Givens rotations, the residual norm, the iteration counter and the convergence
// verdict) lives in a SolverState in HBM.  Kernels read them from there, the last pass of
// each reduction is followed by a one-thread "step" that evaluates the reference's scalar
// statements (safe_divide, sqrt, sym_ortho, the convergence rule) on the device.  The host
// never waits for a scalar: it enqueues iterations ahead and looks at a pinned copy of the
// state `check_lag` iterations behind; once the device has set `done`, every later kernel
// returns at its first instruction, so the result is exactly the reference's: same
// iteration count, x frozen at the iteration that met the tolerance.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>

#include "common.hpp"
#include "blas1_device.hpp"
#include "wave_device.hpp"
#include "solver_device.hpp"
#include "ticket_device.hpp"
#include "ipc_device.hpp"

namespace storm {

// named slots of SolverState::s
enum Slot {
  S_GAMMA = 0, S_PZ, S_GAMMA_NEW, S_BETA, S_ALPHA,
  S_RHO, S_RTV, S_TR, S_TT, S_OMEGA, S_RR, S_RHO_NEW,  // (TR,TT) and (RR,RHO_NEW) stay adjacent: one all-reduce each
  S_TMP, S_HN,
  S_SCRATCH = 32,
  S_ALPHA_SEEN = 48,  // (+ 1: armed) option ticket_verify, RCCL BiCGStab: the alpha the early halo of s was formed with
};

enum StepKind {
  STEP_NONE = 0,
  STEP_CG_INIT,    // gamma = <r,r>; begin(sqrt(gamma))
  STEP_CG_RR,      // gamma_new -> beta, gamma; advance(sqrt(gamma))
  STEP_BICG_INIT,  // rho = <rt,r>; begin(sqrt(rho))
  STEP_BICG_ALPHA, // alpha = rho / <rt,v>
  STEP_BICG_OMEGA, // omega = <t,r> / <t,t>
  STEP_BICG_END,   // err = sqrt(<r,r>); beta from rho_new; advance
  STEP_GMRES_BETA0_OUTER, // beta[0] = sqrt(tmp); begin(beta[0])
  STEP_GMRES_BETA0,       // beta[0] = sqrt(tmp)
  STEP_GMRES_HN,          // hn = sqrt(tmp)
};

__device__ void do_step(int kind, SolverState *st, GmresDev g) {
  double *s = st->s;
  switch (kind) {
    case STEP_CG_INIT:  // SolverCg.hpp:82,85
      begin(st, sqrt(s[S_GAMMA]));
      break;
    case STEP_CG_RR: {  // SolverCg.hpp:110-125
      const double gamma_bar = s[S_GAMMA];
      s[S_GAMMA] = s[S_GAMMA_NEW];
      s[S_BETA] = safe_divide(s[S_GAMMA], gamma_bar);
      advance(st, sqrt(s[S_GAMMA]));
    } break;
    case STEP_BICG_INIT:  // SolverBiCgStab.hpp:88-90
      begin(st, sqrt(s[S_RHO]));
      break;
    case STEP_BICG_ALPHA:  // SolverBiCgStab.hpp:139
      s[S_ALPHA] = safe_divide(s[S_RHO], s[S_RTV]);
      break;
    case STEP_BICG_OMEGA:  // SolverBiCgStab.hpp:159-160
      s[S_OMEGA] = safe_divide(s[S_TR], s[S_TT]);
      break;
    case STEP_BICG_END: {  // :164 then, for the next iteration, :116-118
      const double rho_bar = s[S_RHO];
      s[S_RHO] = s[S_RHO_NEW];
      s[S_BETA] = safe_divide(s[S_ALPHA] * s[S_RHO], s[S_OMEGA] * rho_bar);
      advance(st, sqrt(s[S_RR]));
    } break;
    case STEP_GMRES_BETA0_OUTER:  // SolverGmres.hpp:87,90
      g.beta[0] = sqrt(s[S_TMP]);
      s[S_HN] = g.beta[0];
      begin(st, g.beta[0]);
      break;
    case STEP_GMRES_BETA0:  // SolverGmres.hpp:115
      g.beta[0] = sqrt(s[S_TMP]);
      s[S_HN] = g.beta[0];
      break;
    case STEP_GMRES_HN:  // SolverGmres.hpp:161
      s[S_HN] = sqrt(s[S_TMP]);
      break;
    default: break;
  }
}

struct OutSlots {
  double *p[4];
};

// Final pass of up to 4 simultaneous reductions + the scalar step.  On one rank, and on the peer-window transport
// (use_ipc: the block exchanges its sums with the other ranks itself, ipc_device.hpp), that is ONE launch.
__global__ __launch_bounds__(kBlock) void reduce_step_kernel(const double *__restrict__ partials, int nblocks,
                                                             int k, OutSlots out, int step, SolverState *st,
                                                             GmresDev g, bool force, IpcDev w, int use_ipc) {
  // (`done` is the same decision on every rank and the transport's all-reduce epoch is advanced by the device, by the
  //  all-reduces that run: skipping keeps the ranks in step)
  if (!force && st->done) return;
  __shared__ double lds4[4];
  __shared__ double vals[4];
  for (int j = 0; j < k; ++j) {
    const double *p = partials + (int64_t)j * nblocks;
    double v = 0.0;
#pragma unroll 8
    for (int i = threadIdx.x; i < nblocks; i += kBlock) v += p[i];
    const double sum = block_sum256(v, lds4);
    if (threadIdx.x == 0) vals[j] = sum;
  }
  if (use_ipc) ipc_allreduce_block(w, vals, k);
  else __syncthreads();
  if ((int)threadIdx.x < k) *out.p[threadIdx.x] = vals[threadIdx.x];
  __syncthreads();
  if (step != STEP_NONE && threadIdx.x == 0) do_step(step, st, g);
}

// First pass of a two-pass final reduction (used when a kernel left > 4096 partials, e.g. the
// 65 536 per-block partials of a 256^3 SpMV): kStage2 blocks fold the k arrays to kStage2 each.
__global__ __launch_bounds__(kBlock) void reduce_stage1_kernel(const double *__restrict__ partials, int nblocks,
                                                               double *__restrict__ out, const SolverState *st,
                                                               bool force) {
  if (!force && st->done) return;
  __shared__ double lds4[4];
  const int j = blockIdx.y, g = blockIdx.x;
  const int chunk = (nblocks + gridDim.x - 1) / gridDim.x;
  const int i0 = g * chunk, i1 = min(i0 + chunk, nblocks);
  const double *p = partials + (int64_t)j * nblocks;
  double v = 0.0;
  for (int i = i0 + threadIdx.x; i < i1; i += kBlock) v += p[i];
  const double sum = block_sum256(v, lds4);
  if (threadIdx.x == 0) out[j * gridDim.x + g] = sum;
}

// The same first pass, finished in the kernel (ticket_device.hpp): the block that draws the last ticket folds the
// kStage2 block sums and leaves the total in *out -- the consumer (cg_r_kernel) reads one scalar instead of folding
// kStage2 partials in every one of its 8 192 blocks before its first load.
// use_ipc (peer-window transport): the finishing wave also exchanges the sum with the other ranks (ipc_allreduce_wave).
__global__ __launch_bounds__(kBlock) void reduce_stage1_ticket_kernel(const double *__restrict__ partials, int nblocks,
                                                                      double *__restrict__ out, const SolverState *st,
                                                                      TicketArgs tickets, IpcDev w, int use_ipc) {
  if (st->done) return;
  __shared__ double lds4[4];
  const int g = blockIdx.x;
  const int chunk = (nblocks + gridDim.x - 1) / gridDim.x;
  const int i0 = g * chunk, i1 = min(i0 + chunk, nblocks);
  double v = 0.0;
  for (int i = i0 + threadIdx.x; i < i1; i += kBlock) v += partials[i];
  const double sum = block_sum256(v, lds4);
  if (threadIdx.x >= kWave) return;
  const double mine[1] = {sum};
  double total[1];
  if (ticket_reduce_wave0<1>(tickets, mine, 1, (unsigned)g, gridDim.x, total)) {
    if (use_ipc) ipc_allreduce_wave<1>(w, total, 1);
    if (threadIdx.x == 0) *out = total[0];
  }
}

// ... two sums at once (BiCGStab's <t,s>, <t,t> on the peer-window transport): partials[j * nblocks + i], results in out0 / out1.
__global__ __launch_bounds__(kBlock) void reduce_stage1_ticket2_kernel(const double *__restrict__ partials, int nblocks,
                                                                       double *__restrict__ out0, double *__restrict__ out1,
                                                                       const SolverState *st, TicketArgs tickets, IpcDev w, int use_ipc) {
  if (st->done) return;
  __shared__ double lds4[4];
  const int g = blockIdx.x;
  const int chunk = (nblocks + gridDim.x - 1) / gridDim.x;
  const int i0 = g * chunk, i1 = min(i0 + chunk, nblocks);
  double v0 = 0.0, v1 = 0.0;
  for (int i = i0 + threadIdx.x; i < i1; i += kBlock) v0 += partials[i], v1 += partials[nblocks + i];
  const double s0 = block_sum256(v0, lds4);
  const double s1 = block_sum256(v1, lds4);
  if (threadIdx.x >= kWave) return;
  const double mine[2] = {s0, s1};
  double total[2];
  if (ticket_reduce_wave0<2>(tickets, mine, 2, (unsigned)g, gridDim.x, total)) {
    if (use_ipc) ipc_allreduce_wave<2>(w, total, 2);
    if (threadIdx.x == 0) *out0 = total[0], *out1 = total[1];
  }
}

// Option ticket_verify: sums[j] (a reduction recomputed by the two-launch path) against the slab slots the in-kernel
// reduction filled -- they differ by rounding only (another folding order); anything else raises the sticky flag.
// `before`: the sums were taken of the vectors as they are NOW while the step has already run (iteration count).
struct VerifySlots {
  const double *sum[2];
  const double *slot[2];
  int k;
};
__global__ void verify_kernel(VerifySlots v, SolverState *st, long long iteration_of_slots) {
  // (past convergence the ticketed kernel returned early: nothing to compare)
  if (st->iteration != iteration_of_slots) return;
  for (int j = 0; j < v.k; ++j) {
    const double a = *v.sum[j], b = *v.slot[j];
    double scale = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
    // (the second sum of a pair -- <rt, r> beside <r, r> -- has terms of both signs: its rounding error scales with the
    //  first, not with its own value; a lost block's partial is ~1/blocks of the sum, far above either bound)
    if (j == 1 && fabs(*v.sum[0]) > scale) scale = fabs(*v.sum[0]);
    if (!(fabs(a - b) <= (j == 0 ? 1e-10 : 1e-7) * scale)) st->verify_failed = 1;  // (also catches NaN)
  }
}

__global__ void step_kernel(int step, SolverState *st, GmresDev g, bool force) {
  if (!force && st->done) return;
  do_step(step, st, g);
}

// ---- fused vector kernels ------------------------------------------------------------------------
// Same streaming shape as blas1.hip: one trip per thread, kUnroll x 16 bytes per stream in flight.
static inline int vec_blocks(const storm_hip_ctx *, int64_t n) { return stream_blocks(n); }

#define STORM_STREAM_FOR(base, n2) \
  for (int64_t base = (int64_t)blockIdx.x * (kBlock * kUnroll) + threadIdx.x; base < (n2); \
       base += (int64_t)gridDim.x * (kBlock * kUnroll))

// r <<= b - r (Operator.hpp:98); p <<= r (SolverCg.hpp:81 / SolverBiCgStab.hpp:87 for rt);
// partial <r, r>.
__global__ __launch_bounds__(kBlock) void init_residual_kernel(int64_t n, double *__restrict__ r,
                                                               const double *__restrict__ b,
                                                               double *__restrict__ copy_to,
                                                               double *__restrict__ partials, int nt) {
  __shared__ double lds4[4];
  double acc = 0.0;
  const int64_t n2 = n >> 1;
  double2v *r2 = reinterpret_cast<double2v *>(r), *c2 = reinterpret_cast<double2v *>(copy_to);
  const double2v *b2 = reinterpret_cast<const double2v *>(b);
  nt_dispatch(nt, [&](auto nt) {
  STORM_STREAM_FOR(base, n2) {
    double2v vb[kUnroll], vr[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) vb[u] = ldv(b2 + i, nt), vr[u] = ldv(r2 + i, nt);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        const double2v v = vb[u] - vr[u];
        stv(r2 + i, v, nt);
        if (copy_to) stv(c2 + i, v, nt);
        acc += v.x * v.x;
        acc += v.y * v.y;
      }
    }
  }
  });
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double v = b[n - 1] - r[n - 1];
    r[n - 1] = v;
    if (copy_to) copy_to[n - 1] = v;
    acc += v * v;
  }
  const double s = block_sum256(acc, lds4);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// CG iteration, vector part, in two kernels around the <r,r> reduction (SolverCg.hpp:97-123):
//   cg_r_kernel : alpha = safe_divide(gamma, <p,z>);  r -= alpha z;  partial <r,r>
//   cg_xp_kernel: x += alpha p;  p = r + beta p
// The reference's `x += alpha p` (:98) is moved behind the reduction next to the p update, where p
// is read anyway: same values, 64N instead of 72N bytes per iteration.  cg_xp must apply the x
// update of the iteration in which the solver converged (p no longer matters then), so it is
// keyed on the iteration counter, not on `done`.
// With pz_partials != null the kernel folds the (first-pass) partials of <p,z> itself -- every block the same
// n_pz values in the same order, hence the same alpha -- and the separate final-pass launch disappears.
__global__ __launch_bounds__(kBlock) void cg_r_kernel(int64_t n, SolverState *st, double *__restrict__ r,
                                                      const double *__restrict__ z,
                                                      double *__restrict__ partials, int nt,
                                                      const double *__restrict__ pz_partials, int n_pz, int reverse,
                                                      TicketArgs tickets, IpcDev w, int use_ipc) {
  if (st->done) return;
  __shared__ double lds4[4];
  // `reverse`: the blocks sweep the rows from the far end (see the sweep-direction note in storm_hip_solve_cg);
  // block bx still owns the same rows and the same partial, whichever way the grid is dealt out
  const unsigned bx = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  double pz;
  if (pz_partials) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n_pz; i += kBlock) v += pz_partials[i];
    pz = block_sum256(v, lds4);
    if (blockIdx.x == 0 && threadIdx.x == 0) st->s[S_PZ] = pz;
  } else {
    pz = st->s[S_PZ];
  }
  const double alpha = safe_divide(st->s[S_GAMMA], pz);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->s[S_ALPHA] = alpha;  // for cg_xp_kernel of this iteration
  double acc = 0.0;
  const int64_t n2 = n >> 1;
  double2v *r2 = reinterpret_cast<double2v *>(r);
  const double2v *z2 = reinterpret_cast<const double2v *>(z);
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * kUnroll) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * kUnroll)) {
    double2v vr[kUnroll], vz[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) vr[u] = ldv(r2 + i, nt), vz[u] = ldv(z2 + i, nt);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        vr[u] -= alpha * vz[u];
        stv(r2 + i, vr[u], nt);
        acc += vr[u].x * vr[u].x;
        acc += vr[u].y * vr[u].y;
      }
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
    const double vr = r[n - 1] - alpha * z[n - 1];
    r[n - 1] = vr;
    acc += vr * vr;
  }
  const double s = block_sum256(acc, lds4);
  if (tickets.cnt == nullptr) {
    if (threadIdx.x == 0) partials[bx] = s;
    return;
  }
  // <r, r> finishes here (ticket_device.hpp); the last block runs the scalar step of SolverCg.hpp:110-125 and the
  // convergence rule: no final-pass launch
  if (threadIdx.x >= kWave) return;
  const double mine[1] = {s};
  double total[1];
  if (ticket_reduce_wave0<1>(tickets, mine, 1, bx, gridDim.x, total)) {
    if (use_ipc == 1) ipc_allreduce_wave<1>(w, total, 1);  // the global <r,r>: the same bits on every rank
    if (threadIdx.x == 0) {
      st->s[S_GAMMA_NEW] = total[0];
      // (use_ipc == 2: this rank's sum only -- the host enqueues the all-reduce and the step behind this kernel)
      if (use_ipc != 2) do_step(STEP_CG_RR, st, GmresDev{});
    }
  }
}

// Five streams (3 loads, 2 stores): measured best with ONE 16-byte access per stream and thread in flight
// (tools/cg_kernels_bench.hip at 256^3: U = 1 105.6 us, U = 2 108.2, U = 4 111.4 -- and U = 8 615 us: a wave that
// holds too many loads in flight stalls the memory pipeline), unlike the 2- and 3-stream kernels (U = 4).
constexpr int kUnrollXp = 1;
__global__ __launch_bounds__(kBlock) void cg_xp_kernel(int64_t n, const SolverState *st, long long my_iteration,
                                                       double *__restrict__ x, double *__restrict__ p,
                                                       const double *__restrict__ r, int nt, int reverse) {
  if (st->iteration < my_iteration) return;  // enqueued past convergence: this iteration never ran
  const unsigned bx = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const bool update_p = !st->done && r != nullptr;  // (r == null: the tail of the fused loop -- only x is left to update)
  const double alpha = st->s[S_ALPHA], beta = st->s[S_BETA];
  const int64_t n2 = n >> 1;
  double2v *x2 = reinterpret_cast<double2v *>(x), *p2 = reinterpret_cast<double2v *>(p);
  const double2v *r2 = reinterpret_cast<const double2v *>(r);
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * kUnrollXp) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * kUnrollXp)) {
    double2v vx[kUnrollXp], vp[kUnrollXp], vr[kUnrollXp];
#pragma unroll
    for (int u = 0; u < kUnrollXp; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        vx[u] = ldv(x2 + i, nt), vp[u] = ldv(p2 + i, nt);
        if (update_p) vr[u] = ldv(r2 + i, nt);
      }
    }
#pragma unroll
    for (int u = 0; u < kUnrollXp; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        vx[u] += alpha * vp[u];
        stv(x2 + i, vx[u], nt);
        if (update_p) stv(p2 + i, vr[u] + beta * vp[u], nt);
      }
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    x[i] += alpha * p[i];
    if (update_p) p[i] = r[i] + beta * p[i];
  }
}
static inline int xp_blocks(int64_t n) {
  int64_t b = ((n >> 1) + kBlock * kUnrollXp - 1) / (kBlock * kUnrollXp);
  return (int)(b < 1 ? 1 : (b > 131072 ? 131072 : b));
}

// The two half-steps of a BiCGStab iteration (SolverBiCgStab.hpp:140-141 and :161-164).
//   FIRST : r -= alpha v.  The reference's  x += alpha p  is deferred: nothing reads x before the
//           second half-step, and doing it there saves one read + write of x per iteration.
//   SECOND: x = (x + alpha p) + omega r  (the same two roundings, in the reference's order),
//           r -= omega t, partials of <r,r> and <rt,r>.
template <bool SECOND>
__global__ __launch_bounds__(kBlock) void bicg_update_kernel(int64_t n, SolverState *st, double *__restrict__ x,
                                                             double *__restrict__ r, const double *__restrict__ p,
                                                             const double *__restrict__ w,
                                                             const double *__restrict__ rt,
                                                             double *__restrict__ partials, int nt, int reverse,
                                                             TicketArgs tickets, const double *r_in = nullptr,
                                                             IpcDev ipc_w = IpcDev{}, int use_ipc = 0) {
  // r_in (second half-step): the vector r is READ from (s = r - alpha v, where the apply formed it into a vector of its
  // own); null: r itself
  if (st->done) return;
  const unsigned bx = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;  // the same rows and slots, dealt out from the far end
  __shared__ double lds4[4];
  // With tickets the SpMV before this kernel left the finished sums in the slab and no step kernel ran: every
  // block forms alpha (first half-step, :139) / omega (second, :159-160) itself, block 0 keeps it for later readers.
  double alpha = st->s[S_ALPHA], omega = st->s[S_OMEGA];
  if (tickets.cnt != nullptr) {
    if (!SECOND) {
      alpha = safe_divide(st->s[S_RHO], st->s[S_RTV]);
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->s[S_ALPHA] = alpha;
        // Option ticket_verify over RCCL: the halo of s left BEFORE this kernel, its rows formed by halo_pack_bicg_kernel
        // with an alpha of its own division -- the bits of what the neighbours received depend on it being THIS alpha.
        if (st->s[S_ALPHA_SEEN + 1] != 0.0) {
          if (__double_as_longlong(st->s[S_ALPHA_SEEN]) != __double_as_longlong(alpha)) st->verify_failed = 1;
          st->s[S_ALPHA_SEEN + 1] = 0.0;
        }
      }
    } else {
      omega = safe_divide(st->s[S_TR], st->s[S_TT]);
      if (blockIdx.x == 0 && threadIdx.x == 0) st->s[S_OMEGA] = omega;
    }
  }
  double acc_rr = 0.0, acc_rho = 0.0;
  const int64_t n2 = n >> 1;
  double2v *x2 = reinterpret_cast<double2v *>(x), *r2 = reinterpret_cast<double2v *>(r);
  const double2v *ri2 = r_in ? reinterpret_cast<const double2v *>(r_in) : r2;
  const double2v *p2 = reinterpret_cast<const double2v *>(p), *w2 = reinterpret_cast<const double2v *>(w);
  const double2v *rt2 = reinterpret_cast<const double2v *>(rt);
  constexpr int U = SECOND ? 1 : kUnroll;  // 7 streams: one access per stream in flight (see cg_xp_kernel)
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * U) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * U)) {
    double2v vx[U], vr[U], vw[U], vp[U], vt[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t i = base + q * kBlock;
      if (i < n2) {
        vr[q] = ldv(ri2 + i, nt), vw[q] = ldv(w2 + i, nt);
        if (SECOND) vx[q] = ldv(x2 + i, nt), vp[q] = ldv(p2 + i, nt), vt[q] = ldv(rt2 + i, nt);
      }
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t i = base + q * kBlock;
      if (i < n2) {
        if (!SECOND) {
          vr[q] -= alpha * vw[q];
          stv(r2 + i, vr[q], nt);
        } else {
          vx[q] += alpha * vp[q];
          vx[q] += omega * vr[q];
          vr[q] -= omega * vw[q];
          stv(x2 + i, vx[q], nt);
          stv(r2 + i, vr[q], nt);
          acc_rr += vr[q].x * vr[q].x;
          acc_rr += vr[q].y * vr[q].y;
          acc_rho += vt[q].x * vr[q].x;
          acc_rho += vt[q].y * vr[q].y;
        }
      }
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    if (!SECOND) {
      r[i] -= alpha * w[i];
    } else {
      const double ri = r_in ? r_in[i] : r[i];
      double vx = x[i] + alpha * p[i];
      vx += omega * ri;
      const double vr = ri - omega * w[i];
      x[i] = vx, r[i] = vr;
      acc_rr += vr * vr;
      acc_rho += rt[i] * vr;
    }
  }
  if (SECOND) {
    const double s0 = block_sum256(acc_rr, lds4);
    const double s1 = block_sum256(acc_rho, lds4);
    if (tickets.cnt == nullptr) {
      if (threadIdx.x == 0) partials[bx] = s0, partials[gridDim.x + bx] = s1;
      return;
    }
    if (threadIdx.x >= kWave) return;
    const double mine[2] = {s0, s1};
    double total[2];
    if (ticket_reduce_wave0<2>(tickets, mine, 2, bx, gridDim.x, total)) {
      if (use_ipc == 1) ipc_allreduce_wave<2>(ipc_w, total, 2);  // (peer windows: the global sums, the same bits on every rank)
      if (threadIdx.x == 0) {
        st->s[S_RR] = total[0], st->s[S_RHO_NEW] = total[1];
        st->s[S_OMEGA] = omega;  // (block 0's store of the same value need not be visible to this block yet)
        // (use_ipc == 2, RCCL: this rank's sums only -- the host enqueues the all-reduce and the step behind this kernel)
        if (use_ipc != 2) do_step(STEP_BICG_END, st, GmresDev{});  // :164, :116-118 and the convergence rule
      }
    }
  }
}

// One modified-Gram-Schmidt step fused with the next reduction (SolverGmres.hpp:157-161):
//   w -= h * qa;  partial <w, qb>   (qb == nullptr: partial <w, w>, the norm of :161)
// Same values as the reference's dot -> axpy -> dot chain, one pass over w instead of two.
// h comes either from memory (*h) or, on a single rank with few blocks, from the previous kernel's
// per-block partials: every block folds them itself in the same fixed order (so all blocks hold the
// same h) and block 0 stores it into the Hessenberg -- the separate final-reduction launch between two
// steps disappears, which is what a 128^3 problem (launch-bound MGS chain) is made of.
__global__ __launch_bounds__(kBlock) void mgs_step_kernel(int64_t n, const int *done, double *__restrict__ w,
                                                          const double *h, const double *__restrict__ in_partials,
                                                          int n_in, double *h_store,
                                                          const double *__restrict__ qa,
                                                          const double *qb, double *__restrict__ partials, int nt) {
  if (done && *done) return;
  __shared__ double lds4[4];
  double hv;
  if (in_partials) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n_in; i += kBlock) v += in_partials[i];
    hv = block_sum256(v, lds4);
    if (blockIdx.x == 0 && threadIdx.x == 0) *h_store = hv;
  } else {
    hv = *h;
  }
  double acc = 0.0;
  const int64_t n2 = n >> 1;
  double2v *w2 = reinterpret_cast<double2v *>(w);
  const double2v *a2 = reinterpret_cast<const double2v *>(qa), *b2 = reinterpret_cast<const double2v *>(qb);
  nt_dispatch(nt, [&](auto nt) {
  STORM_STREAM_FOR(base, n2) {
    double2v vw[kUnroll], va[kUnroll], vb[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        vw[u] = ldv(w2 + i, nt), va[u] = ldv(a2 + i, nt);
        if (qb) vb[u] = ldv(b2 + i, nt);
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        vw[u] -= hv * va[u];
        stv(w2 + i, vw[u], nt);
        const double2v o = qb ? vb[u] : vw[u];
        acc += vw[u].x * o.x;
        acc += vw[u].y * o.y;
      }
    }
  }
  });
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double v = w[n - 1] - hv * qa[n - 1];
    w[n - 1] = v;
    acc += v * (qb ? qb[n - 1] : v);
  }
  const double s = block_sum256(acc, lds4);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// GMRES: Givens update of column k and the beta recurrence, SolverGmres.hpp:176-191.
// One wavefront copies column k and the rotations into LDS, lane 0 runs the reference's loop there (the arithmetic of
// gmres_givens_update, solver_device.hpp, in its order: the same bits) -- the k dependent steps cost an LDS access each
// instead of three trips to memory (9.7 -> ~3 us per inner iteration at k ~ 15).
__global__ __launch_bounds__(kWave) void gmres_givens_kernel(SolverState *st, GmresDev g, int k) {
  if (st->done) return;
  __shared__ double hcol[kMaxMulti + 2], cs_sh[kMaxMulti], sn_sh[kMaxMulti];
  const int lane = (int)threadIdx.x, m = g.m;
  for (int i = lane; i <= k; i += kWave) hcol[i] = g.H[(int64_t)i * m + k];
  for (int i = lane; i < k; i += kWave) cs_sh[i] = g.cs[i], sn_sh[i] = g.sn[i];
  __syncthreads();
  if (lane != 0) return;
  hcol[k + 1] = st->s[S_HN];
  for (int i = 0; i < k; ++i) {
    const double chi = cs_sh[i] * hcol[i] + sn_sh[i] * hcol[i + 1];
    hcol[i + 1] = -sn_sh[i] * hcol[i] + cs_sh[i] * hcol[i + 1];
    hcol[i] = chi;
  }
  const double ha = hcol[k], hb = hcol[k + 1];
  const double rr = hypot(ha, hb);
  double cs, sn;
  if (rr > 0.0) cs = ha / rr, sn = hb / rr;
  else cs = 1.0, sn = 0.0;
  g.cs[k] = cs, g.sn[k] = sn;
  hcol[k] = cs * ha + sn * hb;
  hcol[k + 1] = 0.0;
  for (int i = 0; i <= k + 1; ++i) g.H[(int64_t)i * m + k] = hcol[i];
  const double bk = g.beta[k];
  g.beta[k + 1] = -sn * bk;
  g.beta[k] = bk * cs;
  advance(st, fabs(-sn * bk));
}

// TWO modified-Gram-Schmidt steps per pass, the reductions finished in the kernel (ticket_device.hpp):
//   w -= ha qa;  w -= hb qb;                       (ha, hb from the Hessenberg; qa == nullptr: nothing to subtract,
//                                                   qb == nullptr: one vector)
//   then, of the updated w:  a = <w, qc>, b = <w, qd>, c = <qc, qd>   ->   *out_c = a,  *out_d = b - a c
//   (the reference's h = <w - a qc, qd>, by bilinearity -- see mgs_chain_kernel, latency.hip);
//   qd == nullptr: *out_c = <w, qc>;  qc == nullptr: *out_c = <w, w> (SolverGmres.hpp:161).
// 24 B/row/step and half a launch per step instead of 32 B/row/step and two launches (mgs_step_kernel + final pass).
__global__ __launch_bounds__(kBlock) void mgs_pair_kernel(int64_t n, const int *done, double *__restrict__ w,
                                                          const double *ha, const double *hb,
                                                          const double *__restrict__ qa, const double *__restrict__ qb,
                                                          const double *__restrict__ qc, const double *__restrict__ qd,
                                                          double *out_c, double *out_d, TicketArgs tickets, int nt) {
  if (done && *done) return;
  __shared__ double lds4[4];
  const double va = qa ? *ha : 0.0, vb = qb ? *hb : 0.0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const int64_t n2 = n >> 1;
  double2v *w2 = reinterpret_cast<double2v *>(w);
  const double2v *a2 = reinterpret_cast<const double2v *>(qa), *b2 = reinterpret_cast<const double2v *>(qb);
  const double2v *c2 = reinterpret_cast<const double2v *>(qc), *d2 = reinterpret_cast<const double2v *>(qd);
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
    double2v vw = ldv(w2 + i, nt), xa = {0.0, 0.0}, xb = {0.0, 0.0}, xc = {0.0, 0.0}, xd = {0.0, 0.0};
    if (qa) xa = ldv(a2 + i, nt);
    if (qb) xb = ldv(b2 + i, nt);
    if (qc) xc = ldv(c2 + i, nt);
    if (qd) xd = ldv(d2 + i, nt);
    if (qa) {
      vw -= va * xa;
      if (qb) vw -= vb * xb;
      stv(w2 + i, vw, nt);
    }
    if (qc) {
      s0 += vw.x * xc.x, s0 += vw.y * xc.y;
      if (qd) s1 += vw.x * xd.x, s1 += vw.y * xd.y, s2 += xc.x * xd.x, s2 += xc.y * xd.y;
    } else {
      s0 += vw.x * vw.x, s0 += vw.y * vw.y;
    }
  }
  });
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double vw = w[i];
    if (qa) {
      vw -= va * qa[i];
      if (qb) vw -= vb * qb[i];
      w[i] = vw;
    }
    if (qc) {
      s0 += vw * qc[i];
      if (qd) s1 += vw * qd[i], s2 += qc[i] * qd[i];
    } else {
      s0 += vw * vw;
    }
  }
  const double mine[3] = {block_sum256(s0, lds4), block_sum256(s1, lds4), block_sum256(s2, lds4)};
  if (threadIdx.x >= kWave) return;
  double total[3];
  if (ticket_reduce_wave0<3>(tickets, mine, qd ? 3 : 1, blockIdx.x, gridDim.x, total) && threadIdx.x == 0) {
    *out_c = total[0];
    if (qd) *out_d = total[1] - total[0] * total[2];
  }
}

// T modified-Gram-Schmidt steps per pass (T = 3, 4): the pair kernel's scheme with more vectors per trip over w.
//   w -= h[0] qa[0]; ... ; w -= h[na-1] qa[na-1]       (the previous pass's coefficients, in the reference's order)
//   then, of the updated w:  c_j = <w, qc_j>,  g_ij = <qc_i, qc_j> (i < j < nc)
//   ->  out[j] = c_j - sum_{i<j} out[i] g_ij           (= <w - sum_{i<j} h_i qc_i, qc_j>, by bilinearity, j ascending)
//   nc == 0: out[0] = <w, w> (SolverGmres.hpp:161).
// 8 (2 + na + nc) B/row per pass: 20 B/row/step at T = 4 against the pair kernel's 24.  Shape: the streaming kernels'
// (stream_blocks(n) blocks, kUnroll trips per thread), one access per stream and trip in flight -- with up to ten streams
// that is as many as the pair kernel's five with two; the block's partials (up to T + T (T - 1) / 2) are folded once, after
// the last trip, through LDS with one barrier, and finished by tickets.
template <int T>
struct MgsMultiArgs {
  const double *h[T];   // coefficients of the vectors to subtract (device, finished by the previous pass)
  const double *qa[T];  // the vectors to subtract
  const double *qc[T];  // the vectors to project on next
  double *out[T];       // where their coefficients go (nc == 0: out[0] = the norm's square)
  int na, nc;
};
template <int T, int TRIPS>
__global__ __launch_bounds__(kBlock) void mgs_multi_kernel(int64_t n, const int *done, double *__restrict__ w, MgsMultiArgs<T> a,
                                                           TicketArgs tickets, int nt) {
  if (done && *done) return;
  constexpr int NG = T * (T - 1) / 2, NV = T + NG;
  __shared__ double lds[4][NV];
  double hv[T];
#pragma unroll
  for (int j = 0; j < T; ++j) hv[j] = j < a.na ? *a.h[j] : 0.0;
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  const int64_t n2 = n >> 1;
  double2v *w2 = reinterpret_cast<double2v *>(w);
  auto fold = [&](double wx, double wy, const double2v (&xc)[T]) {
    if (a.nc == 0) {
      acc[0] += wx * wx, acc[0] += wy * wy;
      return;
    }
    int g = T;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      acc[j] += wx * xc[j].x, acc[j] += wy * xc[j].y;
#pragma unroll
      for (int i = 0; i < j; ++i, ++g) acc[g] += xc[i].x * xc[j].x, acc[g] += xc[i].y * xc[j].y;
    }
  };
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)blockIdx.x * (kBlock * TRIPS) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * TRIPS)) {
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) {
      const int64_t i = base + u * kBlock;
      if (i >= n2) break;
      double2v vw = ldv(w2 + i, nt), xa[T], xc[T];
#pragma unroll
      for (int j = 0; j < T; ++j) {
        xa[j] = double2v{0.0, 0.0}, xc[j] = double2v{0.0, 0.0};
        if (j < a.na) xa[j] = ldv(reinterpret_cast<const double2v *>(a.qa[j]) + i, nt);
        if (j < a.nc) xc[j] = ldv(reinterpret_cast<const double2v *>(a.qc[j]) + i, nt);
      }
      if (a.na > 0) {
#pragma unroll
        for (int j = 0; j < T; ++j)
          if (j < a.na) vw -= hv[j] * xa[j];
        stv(w2 + i, vw, nt);
      }
      fold(vw.x, vw.y, xc);
    }
  }
  });
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double vw = w[i];
    double2v xc[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
      if (j < a.na) vw -= hv[j] * a.qa[j][i];
      xc[j] = double2v{j < a.nc ? a.qc[j][i] : 0.0, 0.0};
    }
    if (a.na > 0) w[i] = vw;
    fold(vw, 0.0, xc);
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const double s = wave_sum_to_lane63(acc[v]);  // (DPP: ten shuffle trees through the LDS crossbar cost ~1 us per wave)
    if (lane == kWave - 1) lds[wave][v] = s;
  }
  __syncthreads();
  if (threadIdx.x >= kWave) return;
  double mine[NV], total[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) mine[v] = (lds[0][v] + lds[1][v]) + (lds[2][v] + lds[3][v]);
  const int nv = a.nc == 0 ? 1 : NV;
  if (ticket_reduce_wave0<NV>(tickets, mine, nv, blockIdx.x, gridDim.x, total) && threadIdx.x == 0) {
    if (a.nc == 0) {
      *a.out[0] = total[0];
      return;
    }
    double hn[T];
    int g = T;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      double v = total[j];
#pragma unroll
      for (int i = 0; i < j; ++i, ++g) v -= hn[i] * total[g];
      hn[j] = v;
      if (j < a.nc) *a.out[j] = v;
    }
  }
}

// Classical Gram-Schmidt x2: H(j0 : j0 + kk, k) = h_pass0 + h_pass1.
__global__ void gmres_cgs2_combine_kernel(const int *done, double *H, int m, int k, int j0, int kk,
                                          const double *scratch) {
  if (done && *done) return;
  for (int i = 0; i < kk; ++i) H[(j0 + i) * m + k] = scratch[i] + scratch[kMaxMulti + i];
}

// GMRES: back substitution, SolverGmres.hpp:207-212.  One wavefront copies the triangle and beta into LDS, lane 0 runs
// the reference's loops there (the same operations in the same order: one accumulator per row, j ascending) and the
// wavefront stores beta back: the ~k^2 / 2 dependent steps cost an LDS access each instead of two trips to memory
// (147 -> 2x us per restart of GMRES(50) at any size: 3 us of every inner iteration).
__global__ __launch_bounds__(kWave) void gmres_backsolve_kernel(SolverState *st, GmresDev g, int k, bool force) {
  if (!force && st->done) return;
  __shared__ double Hs[kMaxMulti * kMaxMulti], bs[kMaxMulti];
  const int m = g.m, n = k + 1, lane = threadIdx.x;
  // (every load of the lane issued before the first is stored: one after the other they cost a trip to memory each --
  //  fifteen of the kernel's 21 us at k = 29; the upper triangle is all the substitution reads)
  for (int r0 = 0; r0 < n; r0 += 8) {
    double hv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) hv[u] = (r0 + u < n && lane < n && lane >= r0 + u) ? g.H[(r0 + u) * m + lane] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (r0 + u < n && lane < n) Hs[(r0 + u) * kMaxMulti + lane] = hv[u];
  }
  if (lane < n) bs[lane] = g.beta[lane];
  __syncthreads();
  if (lane == 0) {
    for (int i = k; i >= 0; --i) {
      double acc = bs[i];
      for (int j = i + 1; j <= k; ++j) acc -= Hs[i * kMaxMulti + j] * bs[j];
      bs[i] = acc / Hs[i * kMaxMulti + i];
    }
  }
  __syncthreads();
  if (lane < n) g.beta[lane] = bs[lane];
}

// ---- host-side driver helpers -------------------------------------------------------------------------
struct Driver {
  storm_hip_ctx *c;
  const storm_hip_op *op;
  double alpha, beta;
  int64_t n;
  SolverState *st;
  const int *done;
  GmresDev g{nullptr, nullptr, nullptr, nullptr, 0};
  int lag;
  double *d_history = nullptr;

  double *slot(int i) const { return &st->s[i]; }

  // partials -> slots (+ all-reduce over ranks) -> scalar step
  int finish(int nblocks, int k, const int *slots, int step, bool force = false) {
    OutSlots out{};
    double *contiguous = slot(slots[0]);
    bool contig = true;
    for (int j = 0; j < k; ++j) {
      out.p[j] = slot(slots[j]);
      contig &= (slots[j] == slots[0] + j);
    }
    return finish_ptrs(nblocks, k, out, contig ? contiguous : nullptr, step, force);
  }
  int finish_ptrs(int nblocks, int k, OutSlots out, double *contiguous, int step, bool force = false) {
    const double *partials = c->d_partials;
    if (nblocks > kSinglePassPartials) {
      hipLaunchKernelGGL(reduce_stage1_kernel, dim3(kStage2, k), dim3(kBlock), 0, c->stream, c->d_partials,
                         nblocks, c->d_partials2, st, force);
      HIP_TRY(hipGetLastError());
      partials = c->d_partials2;
      nblocks = kStage2;
    }
    IpcDev w{};
    const bool ipc = c->comm != nullptr && comm_ipc_next(c, &w);
    if (c->comm == nullptr || ipc) {
      hipLaunchKernelGGL(reduce_step_kernel, dim3(1), dim3(kBlock), 0, c->stream, partials, nblocks, k,
                         out, step, st, g, force, w, (int)ipc);
      HIP_TRY(hipGetLastError());
      return STORM_HIP_OK;
    }
    hipLaunchKernelGGL(reduce_step_kernel, dim3(1), dim3(kBlock), 0, c->stream, partials, nblocks, k, out,
                       (int)STEP_NONE, st, g, force, w, 0);
    HIP_TRY(hipGetLastError());
    if (contiguous) {
      STORM_TRY(comm_allreduce_sum(c, contiguous, k));
    } else {
      for (int j = 0; j < k; ++j) STORM_TRY(comm_allreduce_sum(c, out.p[j], 1));
    }
    if (step != STEP_NONE) {
      hipLaunchKernelGGL(step_kernel, dim3(1), dim3(1), 0, c->stream, step, st, g, force);
      HIP_TRY(hipGetLastError());
    }
    return STORM_HIP_OK;
  }

  // Option ticket_verify: <a, b0> (and <a, b1>) once more by the two-launch path (per-block partials, then one block
  // folds them) into scratch slots, compared on the device with the slab slots an in-kernel reduction filled.
  // Single rank only (the slots then hold local sums).  `iteration_of_slots`: the value of the iteration counter for
  // which the slots are current (the ticketed kernel may have run the scalar step already).
  int verify(const double *a, const double *b0, const double *b1, int slot0, int slot1, long long iteration_of_slots) {
    if (c->comm != nullptr) return STORM_HIP_OK;
    const double *bs[2] = {b0, b1};
    const int k = b1 ? 2 : 1;
    int nbp = 0;
    STORM_TRY(k_multi_dot_partials(c, a, bs, k, n, &nbp, nullptr));
    // (test hook ticket_verify_inject: the recomputation "loses" one block's partial, as a stale read would)
    STORM_TRY(k_reduce_final(c, c->d_partials, c->opt_ticket_verify_inject != 0 && nbp > 1 ? nbp - 1 : nbp, k, slot(S_SCRATCH + 8), nullptr));
    VerifySlots v{{slot(S_SCRATCH + 8), slot(S_SCRATCH + 9)}, {slot(slot0), slot(slot1 >= 0 ? slot1 : slot0)}, k};
    hipLaunchKernelGGL(verify_kernel, dim3(1), dim3(1), 0, c->stream, v, st, iteration_of_slots);
    HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
  }

  // y = A x, optionally with fused <w, y> / <y, y> partials.  Returns nblocks of partials (0 = not fused).
  // out0 / out1 (slab slots; -1: none): where an in-kernel (ticketed) reduction may leave <w,y> / <y,y>;
  // *ticketed tells whether it did -- then there are no partials to finish (*nblocks is still their count).
  struct CgStep {  // the fused CG step of spmv.hip (CgFuseArgs): end iteration my_iteration - 1, then apply to the new p
    long long my_iteration;
    double *x;
    const double *r;
    double *p_out;
  };
  int apply(const double *x, double *y, const double *dot_w, bool dot_yy, int *nblocks, bool predicated = true,
            int out0 = -1, int out1 = -1, int *ticketed = nullptr, const CgStep *cg = nullptr) {
    SpmvDot sd;
    if (cg != nullptr) {
      sd.cg.iteration = &st->iteration, sd.cg.my_iteration = cg->my_iteration;
      sd.cg.ca = slot(S_ALPHA), sd.cg.cb = slot(S_BETA);
      sd.cg.x = cg->x, sd.cg.r = cg->r, sd.cg.p_out = cg->p_out;
    }
    sd.w = dot_w;
    sd.yy = dot_yy;
    sd.partials = c->d_partials;
    sd.nblocks_out = nblocks;
    if (out0 >= 0) sd.out[0] = slot(out0);
    if (out1 >= 0) sd.out[1] = slot(out1);
    sd.ticketed_out = ticketed;
    if (ticketed) *ticketed = 0;
    const bool want = (dot_w != nullptr || dot_yy) && c->opt_fuse_dot != 0;
    if (!want && nblocks) *nblocks = 0;
    return spmv_launch(op, host_scal(alpha), host_scal(beta), x, y, want ? &sd : nullptr,
                       predicated ? done : nullptr);
  }
};

static int prepare_state(Driver &d, const storm_hip_solver_params *p, double *history) {
  comm_forget_prebegun(d.c);
  storm_hip_ctx *c = d.c;
  STORM_REQUIRE(p->num_iterations >= 0, "solve: num_iterations < 0");
  for (int i = 0; i < kStateRing; ++i) c->h_done_ring[i] = 0;  // (the previous solve ended with a stream wait: nothing posts any more)
  if (history) {
    HIP_TRY(hipMalloc(&d.d_history, sizeof(double) * (size_t)(p->num_iterations + 1)));
    HIP_TRY(hipMemsetAsync(d.d_history, 0, sizeof(double) * (size_t)(p->num_iterations + 1), c->stream));
  }
  STORM_TRY(state_init(c, c->d_state, p->absolute_error_tolerance, p->relative_error_tolerance, p->num_iterations, d.d_history,
                       c->d_done_ring));
  d.lag = p->check_lag > 0 ? p->check_lag : 4;
  if (d.lag > kStateRing - 1) d.lag = kStateRing - 1;
  return STORM_HIP_OK;
}

// Post a status snapshot for iteration `it`; returns true in *stop when the snapshot of
// iteration it - lag says the device is done.
static int post_and_poll(Driver &d, int64_t it, bool *stop) {
  storm_hip_ctx *c = d.c;
  STORM_TRY(ring_post(c, c->ev_ring, it));
  *stop = false;
  if (it >= d.lag) STORM_TRY(ring_wait(c, c->ev_ring, c->h_done_ring, it - d.lag, stop));
  return STORM_HIP_OK;
}

static int collect(Driver &d, storm_hip_solver_result *res, double *history, int64_t applies_fn(int64_t, int64_t),
                   int64_t m) {
  storm_hip_ctx *c = d.c;
  STORM_TRY(state_read(c, c->d_state, &c->h_state[0]));
  {  // (a cooperative kernel of this solve -- CG's, a Gram-Schmidt chain -- timed out: the caller re-runs the solve)
    const int st_coop = lat_check_gave_up(c);
    if (st_coop != STORM_HIP_OK) {
      if (d.d_history) (void)hipFree(d.d_history), d.d_history = nullptr;
      return st_coop;
    }
  }
  res->path_fallback = c->coop_fallback;
  if (c->h_state[0].verify_failed) {
    if (d.d_history) (void)hipFree(d.d_history), d.d_history = nullptr;
    STORM_FAIL(STORM_HIP_E_HIP, "ticket_verify: an in-kernel reduction disagreed with its two-launch recomputation "
                                "(a partial sum was not visible to the block that folded it)");
  }
  const SolverState &h = c->h_state[0];
  res->iterations = h.iteration;
  res->absolute_error = h.absolute_error;
  res->relative_error = h.relative_error;
  res->initial_error = h.initial_error;
  res->converged = h.converged;
  res->num_applies = applies_fn(h.iteration, m);
  if (history && d.d_history) {
    HIP_TRY(hipMemcpy(history, d.d_history, sizeof(double) * (size_t)(h.iteration + 1), hipMemcpyDeviceToHost));
  }
  if (d.d_history) (void)hipFree(d.d_history), d.d_history = nullptr;
  return STORM_HIP_OK;
}

static int check_solve_args(const storm_hip_op *op, const storm_hip_vec *b, storm_hip_vec *x,
                            const storm_hip_solver_params *p, storm_hip_solver_result *r) {
  STORM_REQUIRE(op && b && x && p && r, "solve: null argument");
  STORM_REQUIRE(b->ctx == op->ctx && x->ctx == op->ctx, "solve: context mismatch");
  STORM_REQUIRE(b->n_owned == op->n_rows && x->n_owned == op->n_rows, "solve: operator has %lld rows, b %lld, x %lld",
                (long long)op->n_rows, (long long)b->n_owned, (long long)x->n_owned);
  STORM_REQUIRE(x->n_halo >= op->n_halo, "solve: x has %lld halo rows, operator needs %lld", (long long)x->n_halo,
                (long long)op->n_halo);
  STORM_REQUIRE(b != x, "solve: b and x must not alias");
  return STORM_HIP_OK;
}

struct VecPool {  // work vectors: re-assigned (zeroed) on every solve like SolverCg.hpp:57-59
  std::vector<storm_hip_vec *> v;
  ~VecPool() {  // (in reverse: the context's pool is a stack -- the next solve finds every vector in its old role)
    for (size_t i = v.size(); i-- > 0;) storm_hip_vec_destroy(v[i]);
  }
  // zero = false: the solver writes every owned row of these vectors before it reads it (context.hip, vec_create_work)
  int make(const storm_hip_vec *like, int count, bool zero = true) {
    if (!zero) {
      std::vector<storm_hip_vec *> made((size_t)count, nullptr);
      STORM_TRY(vec_create_work_batch(like, count, made.data()));
      v.insert(v.end(), made.begin(), made.end());
      return STORM_HIP_OK;
    }
    for (int i = 0; i < count; ++i) {
      storm_hip_vec *p = nullptr;
      STORM_TRY(storm_hip_vec_create_like(like, &p));
      v.push_back(p);
    }
    return STORM_HIP_OK;
  }
};

// A launch-bound inner loop replayed from a hipGraph: one iteration's kernels are captured once
// (their arguments never change -- every scalar is read from the device slab) and replayed per
// iteration, which cuts the host cost of ~8 launches to one.  Not used with a communicator (RCCL
// calls and the comm-stream fork stay eager) nor while per-launch profiling events are recorded.
struct IterationGraph {
  hipGraphExec_t exec = nullptr;
  ~IterationGraph() {
    if (exec) (void)hipGraphExecDestroy(exec);
  }
  template <class F>
  int capture(storm_hip_ctx *c, int64_t iterations, F &&enqueue) {
    if (c->opt_graph == 0 || c->comm != nullptr || c->opt_profile_spmv != 0 || iterations < 8) return STORM_HIP_OK;
    HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    const int st = enqueue();
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (st != STORM_HIP_OK || e != hipSuccess || graph == nullptr) {  // fall back to eager launches
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      return st != STORM_HIP_OK ? st : STORM_HIP_OK;
    }
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr, (void)hipGetLastError();
    (void)hipGraphDestroy(graph);
    return STORM_HIP_OK;
  }
  template <class F>
  int launch_or(storm_hip_ctx *c, F &&enqueue) {
    if (exec == nullptr) return enqueue();
    HIP_TRY(hipGraphLaunch(exec, c->stream));
    return STORM_HIP_OK;
  }
};

static int64_t applies_cg(int64_t it, int64_t) { return 1 + it; }
static int64_t applies_bicg(int64_t it, int64_t) { return 1 + 2 * it; }
static int64_t applies_gmres(int64_t it, int64_t m) { return 1 + it + (it + m - 1) / m; }

}  // namespace storm

namespace storm {

// Orthogonalise w = q_{k+1} against q_0 .. q_k (SolverGmres.hpp:157-161): H(0..k, k) and <w, w> (into *norm2_out;
// the caller takes the root and normalises).  H is the (m+1) x m row-major device Hessenberg.
//   gram_schmidt == 0: modified Gram-Schmidt with exactly the reference's values -- H(0,k) = <w,q_0>, then every
//     step applies  w -= H(i,k) q_i  and already accumulates the next reduction (<w,q_{i+1}>, or <w,w>);
//   gram_schmidt == 1: classical Gram-Schmidt applied twice (2 multi-dots + 2 multi-axpys, batched reductions);
//     `scratch` = 2 * kMaxMulti doubles for the two passes' coefficients.
// Shared by storm_hip_solve_gmres below and by the general engine (krylov.hip).
// *normalised (nullable) = true when qn has already been divided by its norm (the cooperative chain does that).
int gmres_orthogonalize(storm_hip_ctx *c, int64_t n, const SolverState *st, const int *done, double *qn,
                        const double *const *q, int k, int m, double *H, double *norm2_out, double *scratch,
                        int gram_schmidt, bool *normalised, const MgsGivens *givens, bool *givens_done,
                        const ChainApply *apply) {
  // apply (nullable): qn = beta q[k] + alpha M(q[k]) has NOT been formed yet -- the cooperative chain does it itself where
  // it can (latency.hip: mgs_chain_quad_kernel<S, T, true>), otherwise it is formed here, before anything reads qn
  if (normalised) *normalised = false;
  if (givens_done) *givens_done = false;
  bool applied = false;
  if (gram_schmidt == 0 && n > 0) {  // small enough for registers: the whole chain as one cooperative kernel
    bool taken = false;
    STORM_TRY(gmres_mgs_chain_coop(c, n, done, qn, q, k, m, H, norm2_out, normalised != nullptr, &taken, givens, apply, &applied));
    if (taken && givens != nullptr && givens_done && c->opt_coop_mgs != 2) *givens_done = true;
    if (taken) {
      if (normalised) *normalised = true;
      return STORM_HIP_OK;
    }
  }
  if (apply != nullptr && !applied)
    STORM_TRY(spmv_launch(apply->op, host_scal(apply->alpha), host_scal(apply->beta), apply->x, qn, nullptr, done));
  const int nbv = stream_blocks(n);
  if (n <= 0) {  // an empty rank: zeros (and its share of the all-reduces)
    for (int i = 0; i <= k; ++i) {
      HIP_TRY(hipMemsetAsync(&H[i * m + k], 0, sizeof(double), c->stream));
      if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, &H[i * m + k], 1));
    }
    HIP_TRY(hipMemsetAsync(norm2_out, 0, sizeof(double), c->stream));
    if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, norm2_out, 1));
    return STORM_HIP_OK;
  }
  if (gram_schmidt == 0 && c->comm == nullptr && c->opt_ticket_reduce != 0 && c->opt_coop_mgs_pairs != 0) {
    // two steps per pass (mgs_pair_kernel): ceil((k + 1) / 2) + 1 launches for the k + 1 basis vectors
    const int nbp = (int)std::min<int64_t>(std::max<int64_t>(1, ((n >> 1) + kBlock - 1) / kBlock),
                                           std::min<int64_t>(32768, c->partials_capacity / 3));
    const TicketArgs t{c->d_tickets, c->d_partials, c->d_ticket_sums};
    const int nti = stream_nt(c, n);
    auto h_of = [&](int i) { return &H[(int64_t)i * m + k]; };
    auto launch = [&](int sub, int nsub, int nxt) {  // subtract q[sub .. sub + nsub), then the dots of q[nxt], q[nxt + 1]
      const int nnext = std::min(2, k + 1 - nxt);    // 2: a pair; 1: one vector; 0: the norm
      hipLaunchKernelGGL(mgs_pair_kernel, dim3(nbp), dim3(kBlock), 0, c->stream, n, done, qn,
                         nsub >= 1 ? h_of(sub) : nullptr, nsub >= 2 ? h_of(sub + 1) : nullptr,
                         nsub >= 1 ? q[sub] : nullptr, nsub >= 2 ? q[sub + 1] : nullptr,
                         nnext >= 1 ? q[nxt] : nullptr, nnext >= 2 ? q[nxt + 1] : nullptr,
                         nnext >= 1 ? h_of(nxt) : norm2_out, nnext >= 2 ? h_of(nxt + 1) : (double *)nullptr, t, nti);
    };
    const int64_t steps = c->opt_mgs_steps;
    // (two trips per thread: tools/multi_stream_bench.hip -- 5.47 TB/s against 5.35 with four and 5.45 with one)
    constexpr int kMgsTrips = 2;
    const int nbm = (int)std::min<int64_t>(std::max<int64_t>(1, ((n >> 1) + kBlock * kMgsTrips - 1) / (kBlock * kMgsTrips)), kMaxStreamBlocks);
    if ((steps == 3 || steps == 4) && k >= 2 && 10 * (int64_t)nbm <= c->partials_capacity &&
        10 * (int64_t)((nbm + kTicketGroup - 1) / kTicketGroup) <= 8 * 2048) {
      // three or four steps per pass (mgs_multi_kernel): ceil((k + 1) / T) + 1 launches
      auto go = [&](auto tag, int sub, int nsub, int nxt) {
        constexpr int T = decltype(tag)::value;
        MgsMultiArgs<T> a{};
        a.na = nsub, a.nc = std::max(0, std::min(T, k + 1 - nxt));
        for (int j = 0; j < T; ++j) {
          a.h[j] = h_of(sub + std::min(j, std::max(nsub - 1, 0))), a.qa[j] = q[std::min(sub + j, k)];
          a.qc[j] = q[std::min(nxt + j, k)], a.out[j] = j < a.nc ? h_of(nxt + j) : norm2_out;
        }
        hipLaunchKernelGGL((mgs_multi_kernel<T, kMgsTrips>), dim3(nbm), dim3(kBlock), 0, c->stream, n, done, qn, a, t, nti);
      };
      if (steps == 4) {
        go(std::integral_constant<int, 4>{}, 0, 0, 0);
        for (int i = 0; i <= k; i += 4) go(std::integral_constant<int, 4>{}, i, std::min(4, k + 1 - i), i + 4);
      } else {
        go(std::integral_constant<int, 3>{}, 0, 0, 0);
        for (int i = 0; i <= k; i += 3) go(std::integral_constant<int, 3>{}, i, std::min(3, k + 1 - i), i + 3);
      }
      HIP_TRY(hipGetLastError());
      return STORM_HIP_OK;
    }
    launch(0, 0, 0);
    for (int i = 0; i <= k; i += 2) launch(i, std::min(2, k + 1 - i), i + 2);
    HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
  }
  if (gram_schmidt == 0) {
    const bool fold_in_consumer = c->comm == nullptr && nbv <= 2048 && 2 * (int64_t)nbv <= c->partials_capacity &&
                                  c->opt_fuse_mgs != 0;
    if (fold_in_consumer) {
      // partials ping-pong between two halves of the workspace: step i folds what step i-1 wrote
      double *cur = c->d_partials, *nxt = c->d_partials + nbv;
      STORM_TRY(k_dot_partials(c, qn, q[0], n, cur, nbv, done));
      for (int i = 0; i <= k; ++i) {
        const double *qb = i < k ? q[i + 1] : nullptr;
        hipLaunchKernelGGL(mgs_step_kernel, dim3(nbv), dim3(kBlock), 0, c->stream, n, done, qn,
                           (const double *)nullptr, cur, nbv, &H[i * m + k], q[i], qb, nxt,
                           stream_nt(c, n));
        HIP_TRY(hipGetLastError());
        std::swap(cur, nxt);
      }
      return k_reduce_final(c, cur, nbv, 1, norm2_out, done);  // <w, w>
    }
    {
      const double *bs[1] = {q[0]};
      STORM_TRY(k_multi_dot(c, qn, bs, 1, n, &H[0 * m + k], done));
      if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, &H[0 * m + k], 1));
    }
    for (int i = 0; i <= k; ++i) {
      double *h = &H[i * m + k];
      const double *qb = i < k ? q[i + 1] : nullptr;
      double *out = i < k ? &H[(i + 1) * m + k] : norm2_out;
      hipLaunchKernelGGL(mgs_step_kernel, dim3(nbv), dim3(kBlock), 0, c->stream, n, done, qn, h,
                         (const double *)nullptr, 0, (double *)nullptr, q[i], qb, c->d_partials,
                         stream_nt(c, n));
      HIP_TRY(hipGetLastError());
      STORM_TRY(k_reduce_final(c, c->d_partials, nbv, 1, out, done));
      if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, out, 1));
    }
    return STORM_HIP_OK;
  }
  // classical Gram-Schmidt applied twice; the second pass's coefficients are added to the first's (same span)
  for (int j0 = 0; j0 <= k; j0 += kMaxMulti) {  // restarts longer than one launch is wide: in chunks
    const int kk = std::min(k + 1 - j0, kMaxMulti);
    for (int pass = 0; pass < 2; ++pass) {
      STORM_TRY(k_multi_dot(c, qn, q + j0, kk, n, scratch + pass * kMaxMulti, done));
      if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, scratch + pass * kMaxMulti, kk));
      STORM_TRY(k_multi_axpy(c, qn, scratch + pass * kMaxMulti, -1.0, q + j0, kk, n, done));
    }
    hipLaunchKernelGGL(gmres_cgs2_combine_kernel, dim3(1), dim3(1), 0, c->stream, done, H, m, k, j0, kk, scratch);
    HIP_TRY(hipGetLastError());
  }
  {
    const double *bs[1] = {qn};
    STORM_TRY(k_multi_dot(c, qn, bs, 1, n, norm2_out, done));  // :161
    if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, norm2_out, 1));
  }
  return STORM_HIP_OK;
}

}  // namespace storm

using namespace storm;

extern "C" {

void storm_hip_solver_params_default(storm_hip_solver_params *p) {
  if (!p) return;
  p->num_iterations = 2000;            // Solver.hpp:67
  p->absolute_error_tolerance = 1e-6;  // Solver.hpp:71
  p->relative_error_tolerance = 1e-6;  // Solver.hpp:72
  p->num_inner_iterations = 50;        // Solver.hpp:159
  p->check_lag = 0;
  p->gram_schmidt = 0;
}

}  // extern "C"

namespace {
struct FusedSolveArgs {
  const storm_hip_op *op;
  double alpha, beta;
  const storm_hip_vec *b;
  storm_hip_vec *x;
  const storm_hip_solver_params *params;
  storm_hip_solver_result *result;
  double *history;
  int (*body)(const FusedSolveArgs &);
};
int run_fused_body(void *p) {
  const FusedSolveArgs &a = *static_cast<const FusedSolveArgs *>(p);
  return a.body(a);
}
// A fused solve that may take a cooperative kernel, re-run without them should one give up (latency.hip)
int fused_solve(FusedSolveArgs a) {
  STORM_TRY(check_solve_args(a.op, a.b, a.x, a.params, a.result));
  storm_hip_ctx *c = a.op->ctx;
  HIP_TRY(hipSetDevice(c->device));
  int fb = 0;
  int st = coop_solve_with_fallback(c, a.x, run_fused_body, &a, &fb);
  if (st == STORM_HIP_OK) a.result->path_fallback = fb;
  // a bounded wait of a transport gave up during this solve (a hand-off flag, a peer window): its result is not one
  if (st == STORM_HIP_OK) st = comm_check_error(c);
  return st;
}

int solve_cg_body(const FusedSolveArgs &args) {
  const storm_hip_op *op = args.op;
  const double alpha = args.alpha, beta = args.beta;
  const storm_hip_vec *b = args.b;
  storm_hip_vec *x = args.x;
  const storm_hip_solver_params *params = args.params;
  storm_hip_solver_result *result = args.result;
  double *history = args.history;
  storm_hip_ctx *c = op->ctx;
  HIP_TRY(hipSetDevice(c->device));
  const int64_t n = op->n_rows;
  Driver d{c, op, alpha, beta, n, c->d_state, &c->d_state->done};
  STORM_TRY(prepare_state(d, params, history));
  VecPool pool;
  if (res_eligible(op, false)) {  // a lattice operator that fits the chip's registers: one persistent kernel, a box per block (resident.hip)
    bool taken = false;
    STORM_TRY(res_solve(false, op, alpha, beta, b->d, x->d, nullptr, c->d_state, &taken));
    if (taken) return ++c->n_resident_solves, collect(d, result, history, applies_cg, 0);
  }
  if (cg_latency_eligible(op)) {  // a small operator: the whole solve as one cooperative kernel (latency.hip)
    STORM_TRY(pool.make(x, 2));  // zero-filled: the kernel relies on that for the first direction
    bool taken = false;
    STORM_TRY(cg_latency_solve(op, alpha, beta, b->d, x->d, pool.v[0]->d, pool.v[1]->d, c->d_state, &taken));
    if (taken) return ++c->n_latency_solves, collect(d, result, history, applies_cg, 0);
    // (the cooperative kernel could not be launched: the throughput path below, noted in result->path_fallback)
  }
  ++c->n_throughput_solves;
  const size_t v0 = pool.v.size();
  // (r: the init apply; p: init_residual's copy; z: the first SpMV -- all before any read; the fused step's second p)
  const bool may_fuse_step = c->opt_cg_fuse != 0 && spmv_can_fuse_cg(op);
  STORM_TRY(pool.make(x, may_fuse_step ? 4 : 3, false));
  // (option cg_roles: which of the four work vectors -- consecutive slots of the context's arena -- plays p, r, z and the
  //  second direction vector: the k-th permutation of (0, 1, 2, 3) in lexicographic order; an A/B knob for placement)
  int role[4] = {0, 1, 2, 3};
  for (int64_t k = 0; k < c->opt_cg_roles % 24; ++k) std::next_permutation(role, role + 4);
  if (!may_fuse_step) role[0] = 0, role[1] = 1, role[2] = 2;
  double *p = pool.v[v0 + role[0]]->d, *r = pool.v[v0 + role[1]]->d, *z = pool.v[v0 + role[2]]->d;
  const int nbv = vec_blocks(c, n);

  // init: r = b - A x; p = r; gamma = <r,r>          SolverCg.hpp:75-85
  int nb = 0;
  STORM_TRY(d.apply(x->d, r, nullptr, false, &nb, false));
  hipLaunchKernelGGL(init_residual_kernel, dim3(nbv), dim3(kBlock), 0, c->stream, n, r, b->d, p, c->d_partials,
                     stream_nt(c, n));
  HIP_TRY(hipGetLastError());
  {
    const int slots[1] = {S_GAMMA};
    STORM_TRY(d.finish(nbv, 1, slots, STEP_CG_INIT, true));
  }
  // One iteration's launches (the scalars live in the slab; only the iteration index varies).
  int64_t cur_it = 0;
  // Sweep directions.  The 256 MB Infinity Cache still holds the END of what the previous kernel streamed (a read
  // served from it runs ~18 % faster than from HBM, tools/mall_probe.hip), so every kernel starts where its
  // predecessor stopped: iteration k even -- SpMV forward, cg_r backward, cg_xp forward; k odd -- the mirror image.
  // Blocks keep their rows and their partial slots: the same bits either way.
  const bool sweep = c->opt_sweep_alternate != 0;  // (per rank: with a communicator too)
  const int nt_stream = (int)(stream_nt(c, n) != 0 && !(sweep && c->opt_sweep_alternate == 2));
  // Reductions that finish inside the kernels producing their partials (ticket_device.hpp), one rank: an iteration
  // is then three launches -- SpMV (+ <p,z>), cg_r (+ <r,r>, beta, the convergence rule), cg_xp.
  // (on the peer-window transport too: the block that finishes a reduction exchanges its sum with the other ranks itself)
  IpcDev ipc_w{};
  const bool ipc = c->comm != nullptr && comm_ipc_next(c, &ipc_w);
  const bool tick = c->opt_ticket_reduce != 0 && (c->comm == nullptr || ipc) && nbv <= kTicketGroup * kTicketMaxGroups;
  // (<p,z> inside the SpMV only where it replaces a whole final-pass launch: with more per-wave partials than one
  // pass folds, the first pass + the fold inside cg_r cost what the ticket tail would add to the SpMV)
  const bool tick_spmv = tick && !ipc && (4 * (int64_t)spmv_grid_blocks(op) <= kSinglePassPartials || c->opt_fold_pz == 0);
  // The fused step (one rank, tiled format-4 operator): iteration k's SpMV kernel first ENDS iteration k - 1 --
  // x += alpha p, p' = r + beta p -- on the rows it loads anyway and applies the operator to p': x and p are no longer
  // streamed by a kernel of their own (cg_xp).  p ping-pongs between two vectors (a tile's old p is another tile's
  // halo).  Two launches + the small first pass per iteration; the last iteration's x update runs behind the loop.
  // (on the peer-window transport too: the marching launch also sends p' of the boundary rows, spmv.hip)
  // (on RCCL too, round 4: there the reductions keep their all-reduce between partials and step -- no tickets --, the
  //  step kernel reads alpha, beta and the iteration counter from the slab exactly as cg_xp_kernel does)
  const bool rccl = c->comm != nullptr && comm_is_rccl(c);
  const bool fuse_step = c->opt_cg_fuse != 0 && c->opt_fuse_dot != 0 && spmv_can_fuse_cg(op) &&
                         (rccl ? true : ((c->comm == nullptr || ipc) && tick && !tick_spmv));
  // RCCL: the local sums still finish inside the kernels that produce them (tickets) -- the library all-reduce and the
  // scalar step follow as launches of their own; two small launches per iteration fewer than partials + final pass
  const bool rtick = rccl && fuse_step && c->opt_ticket_reduce != 0 && c->opt_rccl_ticket != 0 &&
                     nbv <= kTicketGroup * kTicketMaxGroups;
  double *p_alt = nullptr;
  if (fuse_step) p_alt = pool.v[v0 + role[3]]->d, ++c->n_cg_fused_steps;
  int64_t last_enqueued = -1;
  auto enqueue_iteration = [&]() -> int {
    const int q = fuse_step ? 0 : sweep ? (int)(cur_it & 1) : 0;  // (fused: the step kernel forward, cg_r backward, always)
    // z = A p, <p,z>                                  SolverCg.hpp:96-97
    c->spmv_reverse = q;
    int pz_done = 0;  // <p,z> finished inside the SpMV kernel (tickets): cg_r reads it from the slab
    int st_apply;
    if (fuse_step && cur_it > 0) {
      const Driver::CgStep step{(long long)cur_it, x->d, r, p_alt};  // ends iteration cur_it - 1 (SolverCg.hpp:98, :123)
      // (<p,z>: per-wave partials for the final pass below -- finishing it inside the marching kernel by tickets was
      //  measured for this loop and dropped; the host loop's fused step, lazy.hip, does finish it there: one launch less
      //  in front of a host wait)
      st_apply = d.apply(p, z, p, false, &nb, true, -1, -1, &pz_done, &step);
      std::swap(p, p_alt);
    } else {
      st_apply = d.apply(p, z, p, false, &nb, true, tick_spmv ? (int)S_PZ : -1, -1, &pz_done);
    }
    last_enqueued = cur_it;
    c->spmv_reverse = 0;
    STORM_TRY(st_apply);
    const double *pz_partials = nullptr;
    if (pz_done) {
    } else if (nb == 0) {  // operator has a CSR tail: separate dot
      const double *bs[1] = {z};
      STORM_TRY(k_multi_dot(c, p, bs, 1, n, d.slot(S_PZ), d.done));
      if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, d.slot(S_PZ), 1));
    } else if (rtick && (int64_t)nb + kStage2 <= c->partials_capacity) {
      hipLaunchKernelGGL(reduce_stage1_ticket_kernel, dim3(kStage2), dim3(kBlock), 0, c->stream, c->d_partials, nb,
                         d.slot(S_PZ), d.st, TicketArgs{c->d_tickets, c->d_partials + nb, c->d_ticket_sums}, ipc_w, 0);
      HIP_TRY(hipGetLastError());
      STORM_TRY(comm_allreduce_sum(c, d.slot(S_PZ), 1));
    } else if (tick && (nb > kSinglePassPartials || ipc) && c->opt_fold_pz != 0 && (int64_t)nb + kStage2 <= c->partials_capacity) {
      // many partials, one rank: ONE small launch folds them and finishes the sum itself (tickets); cg_r_kernel reads
      // <p,z> from the slab and starts streaming at once.  (The block sums go behind the SpMV's partials.)
      hipLaunchKernelGGL(reduce_stage1_ticket_kernel, dim3(kStage2), dim3(kBlock), 0, c->stream, c->d_partials, nb,
                         d.slot(S_PZ), d.st, TicketArgs{c->d_tickets, c->d_partials + nb, c->d_ticket_sums}, ipc_w, (int)ipc);
      HIP_TRY(hipGetLastError());
    } else if (c->comm == nullptr && nb > kSinglePassPartials && c->opt_fold_pz != 0) {
      // ... without tickets: the first pass here, the fold of its kStage2 results inside cg_r_kernel
      hipLaunchKernelGGL(reduce_stage1_kernel, dim3(kStage2, 1), dim3(kBlock), 0, c->stream, c->d_partials, nb,
                         c->d_partials2, d.st, false);
      HIP_TRY(hipGetLastError());
      pz_partials = c->d_partials2;
    } else {
      const int slots[1] = {S_PZ};
      STORM_TRY(d.finish(nb, 1, slots, STEP_NONE));
    }
    // r -= alpha z; gamma = <r,r>                     SolverCg.hpp:97,99,115
    hipLaunchKernelGGL(cg_r_kernel, dim3(nbv), dim3(kBlock), 0, c->stream, n, d.st, r, z, c->d_partials,
                       nt_stream, pz_partials, (int)kStage2, sweep ? 1 - q : 0,
                       (tick || rtick) ? TicketArgs{c->d_tickets, c->d_partials, c->d_ticket_sums} : TicketArgs{nullptr, nullptr, nullptr},
                       ipc_w, rtick ? 2 : (int)(ipc && tick));
    HIP_TRY(hipGetLastError());
    if (rtick) {
      STORM_TRY(comm_allreduce_sum(c, d.slot(S_GAMMA_NEW), 1));
      hipLaunchKernelGGL(step_kernel, dim3(1), dim3(1), 0, c->stream, (int)STEP_CG_RR, d.st, d.g, false);
      HIP_TRY(hipGetLastError());
    } else if (!tick) {
      const int slots[1] = {S_GAMMA_NEW};
      STORM_TRY(d.finish(nbv, 1, slots, STEP_CG_RR));
    }
    if (tick && c->opt_ticket_verify > 0 && cur_it % c->opt_ticket_verify == 0) {
      // <p, z> as the SpMV / first pass left it, <r, r> as cg_r's last block did (it has advanced the counter already)
      STORM_TRY(d.verify(p, z, nullptr, S_PZ, -1, (long long)(cur_it + 1)));
      STORM_TRY(d.verify(r, r, nullptr, S_GAMMA, -1, (long long)(cur_it + 1)));
    }
    if (fuse_step) return STORM_HIP_OK;  // (the next iteration's step kernel, or the tail below, ends this one)
    // (RCCL, no fused step: the halo of the new direction leaves before cg_xp forms it, comm.hip)
    if (rccl && c->opt_rccl_early_halo != 0 && op->halo.n_nbrs > 0)
      STORM_TRY(comm_halo_exchange_begin_formed(op, 2, r, p, nullptr, d.slot(S_BETA), nullptr, p));
    // x += alpha p; p = r + beta p                    SolverCg.hpp:98,123
    hipLaunchKernelGGL(cg_xp_kernel, dim3(xp_blocks(n)), dim3(kBlock), 0, c->stream, n, d.st, (long long)(cur_it + 1), x->d,
                       p, r, nt_stream, q);
    HIP_TRY(hipGetLastError());
    return STORM_HIP_OK;
  };
  for (int64_t it = 0; it < params->num_iterations; ++it) {
    cur_it = it;
    STORM_TRY(enqueue_iteration());
    bool stop = false;
    STORM_TRY(post_and_poll(d, it, &stop));
    if (stop) break;
  }
  if (fuse_step && last_enqueued >= 0) {
    // the x update of the last enqueued iteration (a no-op when that iteration never ran: the step kernel of the
    // iteration behind the converging one has applied it already)
    hipLaunchKernelGGL(cg_xp_kernel, dim3(xp_blocks(n)), dim3(kBlock), 0, c->stream, n, d.st, (long long)(last_enqueued + 1),
                       x->d, p, (const double *)nullptr, nt_stream, 0);
    HIP_TRY(hipGetLastError());
  }
  comm_forget_prebegun(c);
  return collect(d, result, history, applies_cg, 0);
}

int solve_bicgstab_body(const FusedSolveArgs &args) {
  const storm_hip_op *op = args.op;
  const double alpha = args.alpha, beta = args.beta;
  const storm_hip_vec *b = args.b;
  storm_hip_vec *x = args.x;
  const storm_hip_solver_params *params = args.params;
  storm_hip_solver_result *result = args.result;
  double *history = args.history;
  storm_hip_ctx *c = op->ctx;
  HIP_TRY(hipSetDevice(c->device));
  const int64_t n = op->n_rows;
  Driver d{c, op, alpha, beta, n, c->d_state, &c->d_state->done};
  STORM_TRY(prepare_state(d, params, history));
  VecPool pool;
  if (res_eligible(op, true)) {  // (resident.hip)
    bool taken = false;
    STORM_TRY(pool.make(x, 1, false));  // the shadow residual
    STORM_TRY(res_solve(true, op, alpha, beta, b->d, x->d, pool.v[0]->d, c->d_state, &taken));
    if (taken) return ++c->n_resident_solves, collect(d, result, history, applies_bicg, 0);
  }
  if (cg_latency_eligible(op)) {  // a small operator: the whole solve as one cooperative kernel (latency.hip)
    STORM_TRY(pool.make(x, 4));  // zero-filled: the kernel relies on that for the first direction
    double *const work[4] = {pool.v[0]->d, pool.v[1]->d, pool.v[2]->d, pool.v[3]->d};
    bool taken = false;
    STORM_TRY(bicgstab_latency_solve(op, alpha, beta, b->d, x->d, work, c->d_state, &taken));
    if (taken) return ++c->n_latency_solves, collect(d, result, history, applies_bicg, 0);
  }
  ++c->n_throughput_solves;
  const size_t v0 = pool.v.size();
  // (s = r - alpha v formed inside the second apply -- the marching kernel without its x update -- was measured and dropped:
  //  452 against 445 us per iteration at 256^3; profiles/experiments/r08_pruned_experiments.patch)
  STORM_TRY(pool.make(x, 5, false));  // (r, rt: init; p: the copy of iteration 0; v, t: the SpMVs -- all before any read)
  double *p = pool.v[v0]->d, *r = pool.v[v0 + 1]->d, *rt = pool.v[v0 + 2]->d, *t = pool.v[v0 + 3]->d, *v = pool.v[v0 + 4]->d;
  const int nbv = vec_blocks(c, n);
  const int nbv2 = nbv;  // second half-step: one access per stream in flight, four trips per thread
  int nb = 0;

  // init: r = b - A x; rt = r; rho = <rt,r>           SolverBiCgStab.hpp:82-90
  STORM_TRY(d.apply(x->d, r, nullptr, false, &nb, false));
  hipLaunchKernelGGL(init_residual_kernel, dim3(nbv), dim3(kBlock), 0, c->stream, n, r, b->d, rt, c->d_partials,
                     stream_nt(c, n));
  HIP_TRY(hipGetLastError());
  {
    const int slots[1] = {S_RHO};
    STORM_TRY(d.finish(nbv, 1, slots, STEP_BICG_INIT, true));
  }
  // Sweep directions as in storm_hip_solve_cg: every streaming kernel starts at the end of the rows where its
  // predecessor stopped (what the Infinity Cache still holds); blocks keep their rows and partial slots.
  const bool sweep = c->opt_sweep_alternate != 0 && c->opt_graph == 0;
  int dir = 1;
  auto flip = [&]() -> int { return sweep ? (dir ^= 1) : 0; };
  // ... and reductions finished in-kernel (see storm_hip_solve_cg): five launches per iteration instead of eleven.
  const bool tick = c->opt_ticket_reduce != 0 && c->comm == nullptr && nbv <= kTicketGroup * kTicketMaxGroups;
  const TicketArgs no_tickets{nullptr, nullptr, nullptr}, tickets{c->d_tickets, c->d_partials, c->d_ticket_sums};
  // Peer windows: the applies leave per-wave partials; ONE small launch folds them, finishes the sum by tickets and exchanges
  // it with the other ranks (ipc_device.hpp); the update kernels form alpha / omega themselves and the second half-step's
  // last block all-reduces |r|^2, <rt, r> and runs the scalar step -- as on one rank, plus two small launches per iteration.
  IpcDev ipc_w{};
  const bool ipc_tick = c->opt_ticket_reduce != 0 && c->opt_ipc_bicg_ticket != 0 && c->comm != nullptr && comm_ipc_next(c, &ipc_w) &&
                        nbv <= kTicketGroup * kTicketMaxGroups;
  // RCCL: the halo of the vector an update kernel is about to form leaves BEFORE that kernel (comm.hip)
  const bool early_halo = c->comm != nullptr && comm_is_rccl(c) && c->opt_rccl_early_halo != 0 && op->halo.n_nbrs > 0;
  // RCCL (option rccl_ticket): no scalar-step launch behind the all-reduces of <rt, v> and (<t, r>, <t, t>) -- the update
  // kernels (and the kernel that forms the halo of s) form alpha / omega themselves, as on one rank; the second half-step
  // finishes this rank's |r|^2 and <rt, r> by tickets, the all-reduce and the step follow it: four launches less per iteration
  const bool rccl_tick = c->opt_ticket_reduce != 0 && c->opt_rccl_ticket != 0 && c->comm != nullptr && comm_is_rccl(c) &&
                         nbv <= kTicketGroup * kTicketMaxGroups;
  int ticketed = 0;
  auto apply_dir = [&](const double *xin, double *yout, const double *w, bool yy, int out0, int out1) -> int {
    c->spmv_reverse = flip();
    const int st_apply = d.apply(xin, yout, w, yy, &nb, true, tick ? out0 : -1, tick ? out1 : -1, &ticketed);
    c->spmv_reverse = 0;
    return st_apply;
  };
  // Everything of an iteration after the p update (iteration-invariant arguments).
  int64_t bi_it = 0;  // the iteration being enqueued (option ticket_verify)
  auto enqueue_rest = [&]() -> int {
    // v = A p; alpha = rho / <rt,v>                   :137-139
    STORM_TRY(apply_dir(p, v, rt, false, (int)S_RTV, -1));
    bool alpha_in_kernel = ticketed != 0;  // <rt,v> is in the slab; bicg_update forms alpha itself
    if (alpha_in_kernel) {
    } else if (ipc_tick && nb > 0 && (int64_t)nb + kStage2 <= c->partials_capacity) {
      hipLaunchKernelGGL(reduce_stage1_ticket_kernel, dim3(kStage2), dim3(kBlock), 0, c->stream, c->d_partials, nb,
                         d.slot(S_RTV), d.st, TicketArgs{c->d_tickets, c->d_partials + nb, c->d_ticket_sums}, ipc_w, 1);
      HIP_TRY(hipGetLastError());
      alpha_in_kernel = true;
    } else if (rccl_tick && nb > 0) {
      // ONE launch folds the per-wave partials (tickets), the all-reduce follows; alpha is formed by its consumers
      if ((int64_t)nb + kStage2 <= c->partials_capacity) {
        hipLaunchKernelGGL(reduce_stage1_ticket_kernel, dim3(kStage2), dim3(kBlock), 0, c->stream, c->d_partials, nb,
                           d.slot(S_RTV), d.st, TicketArgs{c->d_tickets, c->d_partials + nb, c->d_ticket_sums}, IpcDev{}, 0);
        HIP_TRY(hipGetLastError());
        STORM_TRY(comm_allreduce_sum(c, d.slot(S_RTV), 1));
      } else {
        const int slots[1] = {S_RTV};
        STORM_TRY(d.finish(nb, 1, slots, STEP_NONE));
      }
      alpha_in_kernel = true;
    } else if (nb == 0) {
      const double *bs[1] = {v};
      STORM_TRY(k_multi_dot(c, rt, bs, 1, n, d.slot(S_RTV), d.done));
      if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, d.slot(S_RTV), 1));
      hipLaunchKernelGGL(step_kernel, dim3(1), dim3(1), 0, c->stream, (int)STEP_BICG_ALPHA, d.st, d.g, false);
      HIP_TRY(hipGetLastError());
    } else {
      const int slots[1] = {S_RTV};
      STORM_TRY(d.finish(nb, 1, slots, STEP_BICG_ALPHA));
    }
    {
      // (RCCL: the halo of s leaves now, under this update and the interior rows of the apply)
      //  (alpha not formed yet: the kernel that forms the rows to send divides rho by <rt, v> itself)
      if (early_halo)
        STORM_TRY(comm_halo_exchange_begin_formed(op, 0, r, nullptr, v, alpha_in_kernel ? d.slot(S_RHO) : d.slot(S_ALPHA),
                                                  alpha_in_kernel ? d.slot(S_RTV) : nullptr, r,
                                                  (alpha_in_kernel && c->opt_ticket_verify > 0) ? d.slot(S_ALPHA_SEEN) : nullptr));
      // r -= alpha v   (x += alpha p is applied in the second half-step)      :140-141
      hipLaunchKernelGGL(bicg_update_kernel<false>, dim3(nbv), dim3(kBlock), 0, c->stream, n, d.st, x->d, r, p, v,
                         rt, c->d_partials, stream_nt(c, n), flip(), alpha_in_kernel ? tickets : no_tickets);
      HIP_TRY(hipGetLastError());
      // t = A r; omega = <t,r> / <t,t>                  :158-160
      STORM_TRY(apply_dir(r, t, r, true, (int)S_TR, (int)S_TT));
    }
    bool omega_in_kernel = ticketed != 0, rccl_end = false;
    if (omega_in_kernel) {
    } else if (rccl_tick && nb > 0) {
      if (2 * (int64_t)nb + 2 * kStage2 <= c->partials_capacity) {
        hipLaunchKernelGGL(reduce_stage1_ticket2_kernel, dim3(kStage2), dim3(kBlock), 0, c->stream, c->d_partials, nb,
                           d.slot(S_TR), d.slot(S_TT), d.st, TicketArgs{c->d_tickets, c->d_partials + 2 * (size_t)nb, c->d_ticket_sums},
                           IpcDev{}, 0);
        HIP_TRY(hipGetLastError());
        STORM_TRY(comm_allreduce_sum(c, d.slot(S_TR), 2));
      } else {
        const int slots[2] = {S_TR, S_TT};
        STORM_TRY(d.finish(nb, 2, slots, STEP_NONE));
      }
      omega_in_kernel = rccl_end = true;
    } else if (ipc_tick && nb > 0 && 2 * (int64_t)nb + 2 * kStage2 <= c->partials_capacity) {
      hipLaunchKernelGGL(reduce_stage1_ticket2_kernel, dim3(kStage2), dim3(kBlock), 0, c->stream, c->d_partials, nb,
                         d.slot(S_TR), d.slot(S_TT), d.st, TicketArgs{c->d_tickets, c->d_partials + 2 * (size_t)nb, c->d_ticket_sums},
                         ipc_w, 1);
      HIP_TRY(hipGetLastError());
      omega_in_kernel = true;
    } else if (nb == 0) {
      const double *bs[2] = {r, t};
      STORM_TRY(k_multi_dot(c, t, bs, 2, n, d.slot(S_TR), d.done));
      if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, d.slot(S_TR), 2));
      hipLaunchKernelGGL(step_kernel, dim3(1), dim3(1), 0, c->stream, (int)STEP_BICG_OMEGA, d.st, d.g, false);
      HIP_TRY(hipGetLastError());
    } else {
      const int slots[2] = {S_TR, S_TT};
      STORM_TRY(d.finish(nb, 2, slots, STEP_BICG_OMEGA));
    }
    // x = (x + alpha p) + omega r; r -= omega t; |r|, <rt,r>    :140, :161-164 (+ :116 of the next iteration)
    hipLaunchKernelGGL(bicg_update_kernel<true>, dim3(nbv2), dim3(kBlock), 0, c->stream, n, d.st, x->d, r,
                       p, t, rt, c->d_partials, stream_nt(c, n), flip(), omega_in_kernel ? tickets : no_tickets,
                       (const double *)nullptr, ipc_w, rccl_end ? 2 : (int)(ipc_tick && omega_in_kernel));
    HIP_TRY(hipGetLastError());
    if (!omega_in_kernel) {
      const int slots[2] = {S_RR, S_RHO_NEW};
      STORM_TRY(d.finish(nbv2, 2, slots, STEP_BICG_END));
    } else if (rccl_end) {
      STORM_TRY(comm_allreduce_sum(c, d.slot(S_RR), 2));
      hipLaunchKernelGGL(step_kernel, dim3(1), dim3(1), 0, c->stream, (int)STEP_BICG_END, d.st, d.g, false);
      HIP_TRY(hipGetLastError());
    } else if (c->opt_ticket_verify > 0 && bi_it % c->opt_ticket_verify == 0) {
      // |r|^2 and the next iteration's rho = <rt, r> as the second half-step's last block left them (it has run
      // STEP_BICG_END: rho_new sits in S_RHO, the counter is advanced)
      STORM_TRY(d.verify(r, r, rt, S_RR, S_RHO, (long long)(bi_it + 1)));
    }
    return STORM_HIP_OK;
  };
  auto enqueue_iteration = [&]() -> int {  // iterations >= 1
    // rho, beta were formed by STEP_BICG_END of the previous iteration (same r): :116-119
    if (early_halo) STORM_TRY(comm_halo_exchange_begin_formed(op, 1, r, p, v, d.slot(S_BETA), d.slot(S_OMEGA), p));
    c->stream_reverse = flip();
    const int st_p = k_bicg_p(c, p, r, dev_scal(d.slot(S_BETA)), dev_scal(d.slot(S_OMEGA)), v, n, d.done);
    c->stream_reverse = 0;
    STORM_TRY(st_p);
    return enqueue_rest();
  };
  IterationGraph graph;
  STORM_TRY(graph.capture(c, params->num_iterations - 1, enqueue_iteration));
  for (int64_t it = 0; it < params->num_iterations; ++it) {
    bi_it = it;
    if (it == 0) {
      STORM_TRY(k_copy(c, p, r, n, d.done));  // :114
      STORM_TRY(enqueue_rest());
    } else {
      STORM_TRY(graph.launch_or(c, enqueue_iteration));
    }
    bool stop = false;
    STORM_TRY(post_and_poll(d, it, &stop));
    if (stop) break;
  }
  comm_forget_prebegun(c);
  return collect(d, result, history, applies_bicg, 0);
}

int solve_gmres_body(const FusedSolveArgs &args) {
  const storm_hip_op *op = args.op;
  const double alpha = args.alpha, beta = args.beta;
  const storm_hip_vec *b = args.b;
  storm_hip_vec *x = args.x;
  const storm_hip_solver_params *params = args.params;
  storm_hip_solver_result *result = args.result;
  double *history = args.history;
  storm_hip_ctx *c = op->ctx;
  HIP_TRY(hipSetDevice(c->device));
  const int64_t n = op->n_rows;
  const int m = (int)params->num_inner_iterations;
  STORM_REQUIRE(m >= 1 && m < kMaxMulti, "solve_gmres: num_inner_iterations = %d outside [1, %d)", m, kMaxMulti);
  Driver d{c, op, alpha, beta, n, c->d_state, &c->d_state->done};
  STORM_TRY(prepare_state(d, params, history));
  VecPool pool;
  STORM_TRY(pool.make(x, m + 1, false));  // q_0 .. q_m (SolverGmres.hpp:60-61): q_0 written by start(), q_k+1 by the apply of iteration k
  std::vector<const double *> q(m + 1);
  for (int i = 0; i <= m; ++i) q[i] = pool.v[i]->d;
  // H, beta, cs, sn                                                  SolverGmres.hpp:56-58
  const size_t gm_doubles = (size_t)(m + 1) * m + (m + 1) + m + m;
  if (gm_doubles > c->gmres_capacity) {  // (kept by the context: no allocation, no hipFree -- a device-wide wait -- per solve)
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_gmres) (void)hipFree(c->d_gmres);
    c->d_gmres = nullptr, c->gmres_capacity = 0;
    HIP_TRY(hipMalloc(&c->d_gmres, sizeof(double) * gm_doubles));
    c->gmres_capacity = gm_doubles;
  }
  double *d_gm = c->d_gmres;
  HIP_TRY(hipMemsetAsync(d_gm, 0, sizeof(double) * gm_doubles, c->stream));
  d.g = GmresDev{d_gm, d_gm + (size_t)(m + 1) * m, d_gm + (size_t)(m + 1) * m + (m + 1),
                 d_gm + (size_t)(m + 1) * m + (m + 1) + m, m};
  const int nbv = vec_blocks(c, n);
  int nb = 0;

  // q0 = b - A x; beta0 = |q0|; q0 /= beta0        (outer_init :82-88 and inner_init :110-116)
  auto start = [&](bool outer) -> int {
    double *q0 = const_cast<double *>(q[0]);
    STORM_TRY(d.apply(x->d, q0, nullptr, false, &nb, !outer));
    if (outer) {
      hipLaunchKernelGGL(init_residual_kernel, dim3(nbv), dim3(kBlock), 0, c->stream, n, q0, b->d,
                         (double *)nullptr, c->d_partials, stream_nt(c, n));
      HIP_TRY(hipGetLastError());
      const int slots[1] = {S_TMP};
      STORM_TRY(d.finish(nbv, 1, slots, STEP_GMRES_BETA0_OUTER, true));
      STORM_TRY(k_scale(c, q0, n, dev_scal(d.slot(S_HN)), true, nullptr));
    } else {
      STORM_TRY(k_axpbz(c, q0, host_scal(1.0), b->d, host_scal(-1.0), q0, n, d.done));
      const double *bs[1] = {q0};
      STORM_TRY(k_multi_dot(c, q0, bs, 1, n, d.slot(S_TMP), d.done));
      if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, d.slot(S_TMP), 1));
      hipLaunchKernelGGL(step_kernel, dim3(1), dim3(1), 0, c->stream, (int)STEP_GMRES_BETA0, d.st, d.g, false);
      HIP_TRY(hipGetLastError());
      STORM_TRY(k_scale(c, q0, n, dev_scal(d.slot(S_HN)), true, d.done));
    }
    return STORM_HIP_OK;
  };
  // x += sum_i beta_i q_i after the back substitution        inner_finalize :207-236
  auto finalize = [&](int k, bool force) -> int {
    hipLaunchKernelGGL(gmres_backsolve_kernel, dim3(1), dim3(kWave), 0, c->stream, d.st, d.g, k, force);
    HIP_TRY(hipGetLastError());
    return k_multi_axpy(c, x->d, d.g.beta, 1.0, q.data(), k + 1, n, force ? nullptr : d.done);
  };

  STORM_TRY(start(true));
  for (int64_t it = 0; it < params->num_iterations; ++it) {
    const int k = (int)(it % m);                         // Solver.hpp:239
    if (k == 0) STORM_TRY(start(false));                 // Solver.hpp:240-242
    double *qn = const_cast<double *>(q[k + 1]);
    // SolverGmres.hpp:155 -- by the chain kernel itself where the operator is a format-4 lattice one (one launch per inner
    // iteration, and qn = A q_k never travels through memory); gmres_orthogonalize applies it otherwise
    const ChainApply chain_apply{op, alpha, beta, q[k]};
    const bool defer_apply = params->gram_schmidt == 0 && c->comm == nullptr && c->opt_coop_mgs != 0 && c->opt_coop_mgs_apply != 0 &&
                             op->halo.n_nbrs == 0;
    if (!defer_apply) STORM_TRY(d.apply(q[k], qn, nullptr, false, &nb));
    bool normalised = false, givens_done = false;
    // (the cooperative Gram-Schmidt chain, when it runs, also takes the root, normalises, applies the Givens
    //  rotations and the convergence rule in its last instructions: an inner iteration is then two launches)
    const MgsGivens givens{d.st, d.g.H, d.g.beta, d.g.cs, d.g.sn, d.slot(S_HN)};
    STORM_TRY(gmres_orthogonalize(c, n, d.st, d.done, qn, q.data(), k, m, d.g.H, d.slot(S_TMP), d.slot(S_SCRATCH),
                                  params->gram_schmidt, &normalised, &givens, &givens_done, defer_apply ? &chain_apply : nullptr));
    if (!givens_done) {
      hipLaunchKernelGGL(step_kernel, dim3(1), dim3(1), 0, c->stream, (int)STEP_GMRES_HN, d.st, d.g, false);
      HIP_TRY(hipGetLastError());
      if (!normalised) STORM_TRY(k_scale(c, qn, n, dev_scal(d.slot(S_HN)), true, d.done));      // :162
      hipLaunchKernelGGL(gmres_givens_kernel, dim3(1), dim3(kWave), 0, c->stream, d.st, d.g, k);    // :176-191
      HIP_TRY(hipGetLastError());
    }
    if (k == m - 1) STORM_TRY(finalize(k, false));       // Solver.hpp:244-246
    bool stop = false;
    STORM_TRY(post_and_poll(d, it, &stop));
    if (stop) break;
  }
  // InnerOuterIterativeSolver::finalize, Solver.hpp:250-257.  The in-loop finalize of the very
  // last iteration was skipped by the `done` predicate, so it always runs here.  (When no
  // iterate() ran the reference's finalize divides by H(0,0) = 0; that is not reproduced.)
  STORM_TRY(state_read(c, c->d_state, &c->h_state[0]));
  const int64_t iters = c->h_state[0].iteration;
  if (iters > 0) STORM_TRY(finalize((int)((iters - 1) % m), true));
  return collect(d, result, history, applies_gmres, m);
}
}  // namespace

extern "C" {

int storm_hip_solve_cg(const storm_hip_op *op, double alpha, double beta, const storm_hip_vec *b, storm_hip_vec *x,
                       const storm_hip_solver_params *params, storm_hip_solver_result *result, double *history) {
  if (op) STORM_TRY(lazy_sync(op->ctx));
  return fused_solve(FusedSolveArgs{op, alpha, beta, b, x, params, result, history, &solve_cg_body});
}
int storm_hip_solve_bicgstab(const storm_hip_op *op, double alpha, double beta, const storm_hip_vec *b, storm_hip_vec *x,
                             const storm_hip_solver_params *params, storm_hip_solver_result *result, double *history) {
  if (op) STORM_TRY(lazy_sync(op->ctx));
  return fused_solve(FusedSolveArgs{op, alpha, beta, b, x, params, result, history, &solve_bicgstab_body});
}
int storm_hip_solve_gmres(const storm_hip_op *op, double alpha, double beta, const storm_hip_vec *b, storm_hip_vec *x,
                          const storm_hip_solver_params *params, storm_hip_solver_result *result, double *history) {
  if (op) STORM_TRY(lazy_sync(op->ctx));
  return fused_solve(FusedSolveArgs{op, alpha, beta, b, x, params, result, history, &solve_gmres_body});
}

}  // extern "C"
