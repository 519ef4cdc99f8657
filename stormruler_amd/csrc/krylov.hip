// The general Krylov engine: every solver of the reference's Solvers/ directory, for any operator
// (stencil operator or callback) and any preconditioner side, as a device-resident loop.
//
// What it stands in for (paths relative to the reference root): the bodies of
//   Solvers/SolverCg.hpp:54-126, SolverBiCgStab.hpp:59-165 and :195-367, SolverGmres.hpp:51-249,
//   SolverCgs.hpp:54-172, SolverTfqmr.hpp:41-204, SolverIdrs.hpp:60-281, SolverRichardson.hpp:48-96,
// driven by IterativeSolver::solve / InnerOuterIterativeSolver (Solver.hpp:116-147, :236-257).
//
// How it is built (nothing like the reference's host loops):
//   * A solver is written against a small ENGINE with three kinds of statements --
//       vector statements   y = sum_t c_t v_t   (one streaming kernel; c_t = host number or scalar REGISTER),
//                           y = A(x), y = P(x)  (stencil SpMV, or the caller's callback, which only enqueues),
//       reductions          reg_j = <a, b_j>    (batched partials kernel + one final pass),
//       scalar programs     short lists of micro-ops on the register file (safe_divide, sqrt, fma, compare,
//                           Givens / back-substitution macros, and the convergence rule of Solver.hpp:132-140)
//     The register file lives in HBM.  Scalar micro-ops are collected and ride in the kernel arguments of the final
//     pass of the reduction that precedes them (one launch for "finish the dot, divide, take the root, test for
//     convergence"), so a recurrence never costs the host a round trip.
//   * The host enqueues iterations ahead of the device and polls a pinned `done` ring `check_lag` iterations
//     behind; once the device's rule fires, every later kernel -- including those a callback enqueues, via
//     ctx->api_done -- returns at its first instruction, so x is frozen at exactly the reference's iteration.
//   * Multi-rank: a reduction's partial results are all-reduced (one call per batch) between the final pass and
//     the scalar program.
#include <cmath>
#include <cstring>
#include <initializer_list>
#include <utility>
#include <vector>

#include "common.hpp"
#include "solver_device.hpp"
#include "ipc_device.hpp"
#include "blas1_device.hpp"
#include "ticket_device.hpp"

namespace storm {
namespace kry {

// ---- the scalar machine ---------------------------------------------------------------------------------
enum : uint16_t {
  SC_MOV,       // d = a
  SC_ADD,       // d = a + b
  SC_SUB,       // d = a - b
  SC_MUL,       // d = a * b
  SC_SDIV,      // d = safe_divide(a, b)          Crow/MathUtils.hpp:49-52
  SC_DIV,       // d = a / b
  SC_NEG,       // d = -a
  SC_SQRT,      // d = sqrt(a)
  SC_FMADD,     // d = d + a * b
  SC_FMSUB,     // d = d - a * b
  SC_LT,        // d = a < b ? 1 : 0
  SC_CMOV,      // d = (b != 0) ? a : d
  SC_SYMORTHO,  // (d, d+1, d+2) = (cs, sn, rr) of (a, b)   Crow/MathUtils.hpp:164-179
  SC_BEGIN,     // Solver.hpp:122-128 with initial error a
  SC_ADVANCE,   // Solver.hpp:132-140 with residual norm a
  SC_GIVENS,    // GMRES column a (an integer, not a register): H(a+1, a) = reg b; rotations; d = |beta_{a+1}|
  SC_BACKSOLVE  // GMRES: solve the a x a ... (a + 1) triangular system for beta (a an integer)
};
struct SOp {
  uint16_t op, d, a, b;
};
constexpr int kProgOps = 100;
constexpr int kProgImm = 8;
constexpr uint16_t kImm0 = 0xFFF0;  // operand codes kImm0 + i read imm[i]
struct SProg {
  int n;
  int aux[5];  // GMRES layout: H0, beta0, cs0, sn0, m (register indices)
  double imm[kProgImm];
  SOp ops[kProgOps];
};
struct RedOut {
  int idx[kMaxMulti];
};

__device__ inline void exec_prog(const SProg &p, double *S, SolverState *st) {
  for (int i = 0; i < p.n; ++i) {
    const SOp o = p.ops[i];
#define RD(r) ((r) >= kImm0 ? p.imm[(r) - kImm0] : S[(r)])
    switch (o.op) {
      case SC_MOV: S[o.d] = RD(o.a); break;
      case SC_ADD: S[o.d] = RD(o.a) + RD(o.b); break;
      case SC_SUB: S[o.d] = RD(o.a) - RD(o.b); break;
      case SC_MUL: S[o.d] = RD(o.a) * RD(o.b); break;
      case SC_SDIV: S[o.d] = safe_divide(RD(o.a), RD(o.b)); break;
      case SC_DIV: S[o.d] = RD(o.a) / RD(o.b); break;
      case SC_NEG: S[o.d] = -RD(o.a); break;
      case SC_SQRT: S[o.d] = sqrt(RD(o.a)); break;
      case SC_FMADD: S[o.d] = S[o.d] + RD(o.a) * RD(o.b); break;
      case SC_FMSUB: S[o.d] = S[o.d] - RD(o.a) * RD(o.b); break;
      case SC_LT: S[o.d] = RD(o.a) < RD(o.b) ? 1.0 : 0.0; break;
      case SC_CMOV:
        if (RD(o.b) != 0.0) S[o.d] = RD(o.a);
        break;
      case SC_SYMORTHO: {
        const double a = RD(o.a), b = RD(o.b), rr = hypot(a, b);
        S[o.d] = rr > 0.0 ? a / rr : 1.0;
        S[o.d + 1] = rr > 0.0 ? b / rr : 0.0;
        S[o.d + 2] = rr;
      } break;
      case SC_BEGIN: begin(st, RD(o.a)); break;
      case SC_ADVANCE: advance(st, RD(o.a)); break;
      case SC_GIVENS: {  // SolverGmres.hpp:176-191
        double *H = S + p.aux[0], *beta = S + p.aux[1], *cs = S + p.aux[2], *sn = S + p.aux[3];
        const int m = p.aux[4], k = o.a;
#define H_(i, j) H[(i) * m + (j)]
        H_(k + 1, k) = S[o.b];
        for (int q = 0; q < k; ++q) {
          const double chi = cs[q] * H_(q, k) + sn[q] * H_(q + 1, k);
          H_(q + 1, k) = -sn[q] * H_(q, k) + cs[q] * H_(q + 1, k);
          H_(q, k) = chi;
        }
        const double a = H_(k, k), b = H_(k + 1, k), rr = hypot(a, b);
        const double c1 = rr > 0.0 ? a / rr : 1.0, s1 = rr > 0.0 ? b / rr : 0.0;
        cs[k] = c1, sn[k] = s1;
        H_(k, k) = c1 * H_(k, k) + s1 * H_(k + 1, k);
        H_(k + 1, k) = 0.0;
        beta[k + 1] = -s1 * beta[k];
        beta[k] *= c1;
        S[o.d] = fabs(beta[k + 1]);
      } break;
      case SC_BACKSOLVE: {  // SolverGmres.hpp:207-212
        double *H = S + p.aux[0], *beta = S + p.aux[1];
        const int m = p.aux[4], k = o.a;
        for (int q = k; q >= 0; --q) {
          for (int j = q + 1; j <= k; ++j) beta[q] -= H_(q, j) * beta[j];
          beta[q] /= H_(q, q);
        }
#undef H_
      } break;
      default: break;
    }
#undef RD
  }
}

// Final pass of k simultaneous reductions (out.idx[j] = register of sum j) + the scalar program behind them.  On the
// peer-window transport (use_ipc) the block also exchanges its sums with the other ranks in between: one launch.
__global__ __launch_bounds__(kBlock) void reduce_prog_kernel(const double *__restrict__ partials, int nblocks, int k,
                                                             RedOut out, double *S, SolverState *st, SProg prog,
                                                             const int *done, IpcDev w, int use_ipc) {
  // (`done` is the same decision on every rank, and the transport's all-reduce epoch is advanced by the device, by the
  //  all-reduces that run: skipping here keeps the ranks in step)
  if (done && *done) return;
  __shared__ double lds4[4];
  __shared__ double vals[kMaxMulti];
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
  if ((int)threadIdx.x < k) S[out.idx[threadIdx.x]] = vals[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0 && prog.n > 0) {
    __threadfence();
    exec_prog(prog, S, st);
  }
}

__global__ __launch_bounds__(kBlock) void reduce_stage1_kernel2(const double *__restrict__ partials, int nblocks,
                                                                double *__restrict__ out, const int *done) {
  if (done && *done) return;
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

// A scalar program alone; with nscatter > 0 first S[out.idx[j]] = S[scr + j] (results of an all-reduce).
__global__ void sprog_kernel(double *S, SolverState *st, SProg prog, int nscatter, RedOut out, int scr,
                             const int *done) {
  if (done && *done) return;
  for (int j = 0; j < nscatter; ++j) S[out.idx[j]] = S[scr + j];
  exec_prog(prog, S, st);
}

// o + c * v with ONE rounding per component, spelled out: the same statement must give the same bits whichever
// kernel evaluates it (alone, paired with its successor, with a reduction folded in).
__device__ __forceinline__ double2v fma2(double c, double2v v, double2v o) {
  double2v r;
  r.x = __builtin_fma(c, v.x, o.x), r.y = __builtin_fma(c, v.y, o.y);
  return r;
}
// The `nt` argument of the streaming kernels carries two flags: bit 0 = non-temporal accesses, bit 1 = deal the
// blocks out from the far end of the rows (the engine's sweep-direction scheme; a block keeps its rows and slots).
__device__ __forceinline__ unsigned sweep_block(int flags) { return (flags & 2) ? gridDim.x - 1 - blockIdx.x : blockIdx.x; }

// ---- reductions in ONE launch -----------------------------------------------------------------------------------
// A reduction is "partials kernel, then a one-block final pass that also runs the scalar program": two launches, and
// on the reference's own mesh sizes an iteration is nothing but launches (~4 us each, dependent).  Here the partials
// kernel finishes the job itself (ticket_device.hpp: two levels of tickets, partials published by awaited atomic
// exchange, fixed folding order): the block that draws the last ticket holds the sums, writes the registers and
// runs the scalar program.  The engine holds the partials kernel back until the program behind it is complete.
struct FinalPass {
  int *tickets;   // ticket_device.hpp counters (self re-arming)
  double *part2;  // [k][groups] group sums
  int k;
  RedOut out;
  double *S;
  SolverState *st;
  SProg prog;
};

// `mine[j]`: this block's partial of sum j (the same value in every thread).
template <int KMAX>
__device__ __forceinline__ void publish_and_finish(double *partials, const double (&mine)[KMAX], const FinalPass &f,
                                                   unsigned slot) {
  if (threadIdx.x >= kWave) return;
  double total[KMAX];
  const TicketArgs t{f.tickets, partials, f.part2};
  if (ticket_reduce_wave0<KMAX>(t, mine, f.k, slot, gridDim.x, total) && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j)
      if (j < f.k) f.S[f.out.idx[j]] = total[j];
    if (f.prog.n > 0) exec_prog(f.prog, f.S, f.st);
  }
}

template <int KB>
__global__ __launch_bounds__(kBlock) void dots_prog_kernel(int64_t n, const double *__restrict__ a, DotPtrs bs,
                                                           double *partials, const int *done, int nt, FinalPass f) {
  if (done && *done) return;
  __shared__ double lds4[4];
  double acc[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) acc[j] = 0.0;
  multi_dot_accumulate<KB>(n, a, bs, nt, acc);
  double mine[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) mine[j] = block_sum256(acc[j], lds4);
  publish_and_finish<KB>(partials, mine, f, sweep_block(nt));
}

// ---- vector statements ------------------------------------------------------------------------------------
__device__ __forceinline__ double ld_coef(const Scal &s) { return s.p ? (*s.p) * s.sign : s.v; }

struct LinArgs {
  double *y;
  const double *v[4];
  Scal c[4];
  const double *cond;  // nullable: run only when *cond != 0
};

// 16-byte accesses per stream and thread in flight: 4 for up to three streams, fewer beyond (measured,
// tools/cg_kernels_bench.hip: a 5-stream kernel runs 5 % faster with 1 than with 4, and collapses with 8).
__host__ __device__ constexpr int lin_unroll(int nt) { return nt <= 2 ? 4 : (nt == 3 ? 2 : 1); }

// y = c0 v0 + c1 v1 + ... (left to right), or NESTED (NT = 3):  y = v0 + c1 * (v1 + c2 * v2).
// Operands may alias y (every element is read before it is written by the same lane).
// `gate` (nullable): the statement was issued BEFORE the convergence rule of iteration gate_val - 1 but runs behind
// it (the engine held it back): it must execute iff that rule was evaluated at all -- the iteration counter has reached
// gate_val -- even when the rule then declared the solve done (the x update of the converging iteration).
template <int NT, bool NESTED>
__global__ __launch_bounds__(kBlock) void lin_kernel(int64_t n, LinArgs a, const int *done, int nt,
                                                     const long long *gate, long long gate_val) {
  if (gate ? (*gate < gate_val) : (done && *done)) return;
  if (a.cond && *a.cond == 0.0) return;
  const unsigned bx = sweep_block(nt);
  nt &= 1;
  double c[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) c[t] = ld_coef(a.c[t]);
  const int64_t n2 = n >> 1;
  double2v *y2 = reinterpret_cast<double2v *>(a.y);
  constexpr int U = lin_unroll(NT);
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * U) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * U)) {
    double2v v[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
#pragma unroll
        for (int t = 0; t < NT; ++t) v[u][t] = ldv(reinterpret_cast<const double2v *>(a.v[t]) + i, nt);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        double2v o;
        if constexpr (NESTED) {
          o = v[u][0] + c[1] * (v[u][1] + c[2] * v[u][NT - 1]);
        } else {
          o = c[0] * v[u][0];
#pragma unroll
          for (int t = 1; t < NT; ++t) o = fma2(c[t], v[u][t], o);
        }
        stv(y2 + i, o, nt);
      }
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double o;
    if constexpr (NESTED) {
      o = a.v[0][i] + c[1] * (a.v[1][i] + c[2] * a.v[NT - 1][i]);
    } else {
      o = c[0] * a.v[0][i];
      for (int t = 1; t < NT; ++t) o = __builtin_fma(c[t], a.v[t][i], o);
    }
    a.y[i] = o;
  }
}


// TWO consecutive vector statements in one pass, executed per element in program order (both are elementwise, so
// that is exactly their sequential meaning): y1 = sum c1_t v1_t;  y2 = sum c2_t v2_t, where an operand of the second
// that IS y1 takes the new value.  Every operand is loaded before anything is stored, so the second statement may
// overwrite an operand of the first ("x += alpha p;  p = r + beta p" -- p is read once: 40 instead of 48 bytes per
// row, one launch instead of two).
template <int NT1, int NT2>
__global__ __launch_bounds__(kBlock) void lin2_kernel(int64_t n, LinArgs a1, LinArgs a2, const int *done, int nt,
                                                      const long long *gate, long long gate_val) {
  bool run2 = !(done && *done);
  bool run1 = gate ? (*gate >= gate_val) : run2;  // (see lin_kernel; run2 implies run1)
  if (a1.cond && *a1.cond == 0.0) run1 = false;   // a conditional statement (TFQMR1's `if (omega < tau) x = d`)
  if (a2.cond && *a2.cond == 0.0) run2 = false;
  if (!run1 && !run2) return;
  const unsigned bx = sweep_block(nt);
  nt &= 1;
  double c1[NT1], c2[NT2];
  bool from1[NT2];
#pragma unroll
  for (int t = 0; t < NT1; ++t) c1[t] = ld_coef(a1.c[t]);
#pragma unroll
  for (int t = 0; t < NT2; ++t) c2[t] = ld_coef(a2.c[t]), from1[t] = run1 && a2.v[t] == a1.y;
  const int64_t n2 = n >> 1;
  double2v *y1 = reinterpret_cast<double2v *>(a1.y), *y2 = reinterpret_cast<double2v *>(a2.y);
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t i = (int64_t)bx * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
    double2v v1[NT1], v2[NT2];
#pragma unroll
    for (int t = 0; t < NT1; ++t) v1[t] = ldv(reinterpret_cast<const double2v *>(a1.v[t]) + i, nt);
#pragma unroll
    for (int t = 0; t < NT2; ++t) v2[t] = ldv(reinterpret_cast<const double2v *>(a2.v[t]) + i, nt);
    double2v o1 = c1[0] * v1[0];
#pragma unroll
    for (int t = 1; t < NT1; ++t) o1 = fma2(c1[t], v1[t], o1);
    double2v o2 = c2[0] * (from1[0] ? o1 : v2[0]);
#pragma unroll
    for (int t = 1; t < NT2; ++t) o2 = fma2(c2[t], from1[t] ? o1 : v2[t], o2);
    if (run1) stv(y1 + i, o1, nt);
    if (run2) stv(y2 + i, o2, nt);
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double w1[NT1], w2[NT2];
    for (int t = 0; t < NT1; ++t) w1[t] = a1.v[t][i];
    for (int t = 0; t < NT2; ++t) w2[t] = a2.v[t][i];
    double o1 = c1[0] * w1[0];
    for (int t = 1; t < NT1; ++t) o1 = __builtin_fma(c1[t], w1[t], o1);
    double o2 = c2[0] * (from1[0] ? o1 : w2[0]);
    for (int t = 1; t < NT2; ++t) o2 = __builtin_fma(c2[t], from1[t] ? o1 : w2[t], o2);
    if (run1) a1.y[i] = o1;
    if (run2) a2.y[i] = o2;
  }
}

// The same statement with reductions of its RESULT folded in: per-block partials of <y, y> (dot_yy) and / or
// <y, w> into partials[j * gridDim.x + block] -- "r -= alpha z; gamma = <r, r>" is one pass over r, not two.
// HASW: a second operand w is streamed for <y, w> (it counts as a stream when the accesses in flight are chosen: "r -= alpha z;
// <r, r>" is a three-stream kernel like cg_r and runs with four, 79 -> 6x us at 256^3).
template <int NT, bool HASW>
__device__ __forceinline__ void lin_dot_body(int64_t n, const LinArgs &a, const double *w, int nt, double &acc_yy,
                                             double &acc_yw) {
  const unsigned bx = sweep_block(nt);
  nt &= 1;
  double c[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) c[t] = ld_coef(a.c[t]);
  const int64_t n2 = n >> 1;
  double2v *y2 = reinterpret_cast<double2v *>(a.y);
  const double2v *w2 = reinterpret_cast<const double2v *>(w);
  constexpr int U = lin_unroll(NT + (HASW ? 1 : 0));
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * U) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * U)) {
    double2v v[U][NT], vw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
#pragma unroll
        for (int t = 0; t < NT; ++t) v[u][t] = ldv(reinterpret_cast<const double2v *>(a.v[t]) + i, nt);
        if (w) vw[u] = ldv(w2 + i, nt);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        double2v o = c[0] * v[u][0];
#pragma unroll
        for (int t = 1; t < NT; ++t) o = fma2(c[t], v[u][t], o);
        stv(y2 + i, o, nt);
        acc_yy += o.x * o.x;
        acc_yy += o.y * o.y;
        if (w) acc_yw += o.x * vw[u].x, acc_yw += o.y * vw[u].y;
      }
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double o = c[0] * a.v[0][i];
    for (int t = 1; t < NT; ++t) o = __builtin_fma(c[t], a.v[t][i], o);
    a.y[i] = o;
    acc_yy += o * o;
    if (w) acc_yw += o * w[i];
  }
}

template <int NT, bool HASW>
__global__ __launch_bounds__(kBlock) void lin_dot_kernel(int64_t n, LinArgs a, const double *w, int dot_yy,
                                                         double *__restrict__ partials, const int *done, int nt) {
  if (done && *done) return;
  __shared__ double lds4[4];
  double acc_yy = 0.0, acc_yw = 0.0;
  lin_dot_body<NT, HASW>(n, a, w, nt, acc_yy, acc_yw);
  int j = 0;
  if (dot_yy) {
    const double sum = block_sum256(acc_yy, lds4);
    if (threadIdx.x == 0) partials[(int64_t)j * gridDim.x + sweep_block(nt)] = sum;
    ++j;
  }
  if (w) {
    const double sum = block_sum256(acc_yw, lds4);
    if (threadIdx.x == 0) partials[(int64_t)j * gridDim.x + sweep_block(nt)] = sum;
  }
}

// ... and with the final pass in the last block (see publish_and_finish).
template <int NT, bool HASW>
__global__ __launch_bounds__(kBlock) void lin_dot_prog_kernel(int64_t n, LinArgs a, const double *w, int dot_yy,
                                                              double *partials, const int *done, int nt, FinalPass f) {
  if (done && *done) return;
  __shared__ double lds4[4];
  double acc_yy = 0.0, acc_yw = 0.0;
  lin_dot_body<NT, HASW>(n, a, w, nt, acc_yy, acc_yw);
  double mine[2] = {0.0, 0.0};
  int j = 0;
  if (dot_yy) mine[j++] = block_sum256(acc_yy, lds4);
  if (w) mine[j] = block_sum256(acc_yw, lds4);
  publish_and_finish<2>(partials, mine, f, sweep_block(nt));
}

// A diagonal preconditioner and the reductions behind it in one pass: z = d .* r, <r, z>, <r, r> (preconditioned CG,
// SolverCg.hpp:100-115).  Rows per block and order of a thread's terms are those of multi_dot_accumulate<2>
// (blas1_device.hpp), so the sums carry the bits of the separate vmul + multi-dot.
__global__ __launch_bounds__(kBlock) void vmul_dots_prog_kernel(int64_t n, double *__restrict__ z,
                                                                const double *__restrict__ d,
                                                                const double *__restrict__ r, double *partials,
                                                                const int *done, int nt, FinalPass f) {
  if (done && *done) return;
  __shared__ double lds4[4];
  const unsigned bx = sweep_block(nt);
  const int64_t n2 = n >> 1;
  double2v *z2 = reinterpret_cast<double2v *>(z);
  const double2v *d2 = reinterpret_cast<const double2v *>(d), *r2 = reinterpret_cast<const double2v *>(r);
  double acc_rz = 0.0, acc_rr = 0.0;
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * kUnroll) + threadIdx.x; base < n2; base += (int64_t)gridDim.x * (kBlock * kUnroll)) {
    double2v vd[kUnroll], vr[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) vd[u] = ldv(d2 + i, nt), vr[u] = ldv(r2 + i, nt);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        const double2v vz = vd[u] * vr[u];
        stv(z2 + i, vz, nt);
        acc_rz += vr[u].x * vz.x;
        acc_rz += vr[u].y * vz.y;
        acc_rr += vr[u].x * vr[u].x;
        acc_rr += vr[u].y * vr[u].y;
      }
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
    const double vz = d[n - 1] * r[n - 1];
    z[n - 1] = vz;
    acc_rz += r[n - 1] * vz;
    acc_rr += r[n - 1] * r[n - 1];
  }
  const double mine[2] = {block_sum256(acc_rz, lds4), block_sum256(acc_rr, lds4)};
  publish_and_finish<2>(partials, mine, f, bx);
}

// A held-back vector statement, the statement that follows it, and the reductions of THAT statement's result, in
// one pass (lin2_kernel + lin_dot_prog_kernel): BiCGStab's "x += alpha p + omega s;  r = s - omega t;  |r|^2, <rt, r>"
// reads x, p, r, t, rt and writes x, r once -- the hand-fused loop's second half-step.
template <int NT1, int NT2, bool HASW>
__global__ __launch_bounds__(kBlock) void lin2_dot_prog_kernel(int64_t n, LinArgs a1, LinArgs a2, const double *w,
                                                               int dot_yy, double *partials, const int *done, int nt,
                                                               FinalPass f) {
  if (done && *done) return;
  __shared__ double lds4[4];
  const unsigned bx = sweep_block(nt);
  double c1[NT1], c2[NT2];
  bool from1[NT2];
#pragma unroll
  for (int t = 0; t < NT1; ++t) c1[t] = ld_coef(a1.c[t]);
#pragma unroll
  for (int t = 0; t < NT2; ++t) c2[t] = ld_coef(a2.c[t]), from1[t] = a2.v[t] == a1.y;
  const bool w_from1 = w == a1.y;
  const int64_t n2 = n >> 1;
  double2v *y1 = reinterpret_cast<double2v *>(a1.y), *y2 = reinterpret_cast<double2v *>(a2.y);
  const double2v *w2 = reinterpret_cast<const double2v *>(w);
  double acc_yy = 0.0, acc_yw = 0.0;
  // (the rows of a block and the order of a thread's terms are those of lin_dot_body for the second statement: the
  //  partial sums -- and the reduction's bits -- do not depend on whether a held-back statement rode along)
  constexpr int U = lin_unroll(NT2 + (HASW ? 1 : 0));
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * U) + threadIdx.x; base < n2; base += (int64_t)gridDim.x * (kBlock * U)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kBlock;
      if (i < n2) {
        double2v v1[NT1], v2[NT2], vw = {0.0, 0.0};
#pragma unroll
        for (int t = 0; t < NT1; ++t) v1[t] = ldv(reinterpret_cast<const double2v *>(a1.v[t]) + i, nt);
#pragma unroll
        for (int t = 0; t < NT2; ++t) v2[t] = ldv(reinterpret_cast<const double2v *>(a2.v[t]) + i, nt);
        if (w) vw = ldv(w2 + i, nt);
        double2v o1 = c1[0] * v1[0];
#pragma unroll
        for (int t = 1; t < NT1; ++t) o1 = fma2(c1[t], v1[t], o1);
        double2v o2 = c2[0] * (from1[0] ? o1 : v2[0]);
#pragma unroll
        for (int t = 1; t < NT2; ++t) o2 = fma2(c2[t], from1[t] ? o1 : v2[t], o2);
        stv(y1 + i, o1, nt);
        stv(y2 + i, o2, nt);
        if (w_from1) vw = o1;
        acc_yy += o2.x * o2.x;
        acc_yy += o2.y * o2.y;
        if (w) acc_yw += o2.x * vw.x, acc_yw += o2.y * vw.y;
      }
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double w1[NT1], w2v[NT2];
    for (int t = 0; t < NT1; ++t) w1[t] = a1.v[t][i];
    for (int t = 0; t < NT2; ++t) w2v[t] = a2.v[t][i];
    const double wl = w ? w[i] : 0.0;
    double o1 = c1[0] * w1[0];
    for (int t = 1; t < NT1; ++t) o1 = __builtin_fma(c1[t], w1[t], o1);
    double o2 = c2[0] * (from1[0] ? o1 : w2v[0]);
    for (int t = 1; t < NT2; ++t) o2 = __builtin_fma(c2[t], from1[t] ? o1 : w2v[t], o2);
    a1.y[i] = o1;
    a2.y[i] = o2;
    acc_yy += o2 * o2;
    if (w) acc_yw += o2 * (w_from1 ? o1 : wl);
  }
  double mine[2] = {0.0, 0.0};
  int j = 0;
  if (dot_yy) mine[j++] = block_sum256(acc_yy, lds4);
  if (w) mine[j] = block_sum256(acc_yw, lds4);
  publish_and_finish<2>(partials, mine, f, bx);
}

}  // namespace kry
}  // namespace storm

using namespace storm;
using namespace storm::kry;

// ---- the engine ---------------------------------------------------------------------------------------------
namespace {

struct Coef {  // a coefficient of a vector statement: a host number, or +-register
  int reg;
  double v, sign;
};
inline Coef R(int reg) { return Coef{reg, 0.0, 1.0}; }
inline Coef mR(int reg) { return Coef{reg, 0.0, -1.0}; }
inline Coef num(double v) { return Coef{-1, v, 1.0}; }
typedef storm_hip_vec *V;
struct Term {
  Coef c;
  const storm_hip_vec *v;
};

// registers every method has
enum { R_ZERO = 0, R_ONE, R_ERR, R_T0, R_T1, R_T2, R_T3, R_SCR /* kMaxMulti all-reduce slots */, R_USER = R_SCR + kMaxMulti };

}  // namespace

struct KrylovEngine {
  storm_hip_ctx *c = nullptr;
  int method = 0;
  // operator / preconditioner
  const storm_hip_op *op = nullptr;
  double op_alpha = 0.0, op_beta = 0.0;
  storm_hip_apply_fn op_fn = nullptr;
  void *op_user = nullptr;
  storm_hip_apply_fn pre_fn = nullptr;
  void *pre_user = nullptr;
  const storm_hip_vec *pre_diag = nullptr;
  int side = STORM_HIP_RIGHT;
  double relaxation = 1.0e-4;  // SolverRichardson.hpp:45
  // device state (own: solves may nest, e.g. a solver used as another solver's preconditioner)
  SolverState *d_st = nullptr, *h_st = nullptr;
  unsigned long long *h_ring = nullptr, *d_ring = nullptr;
  std::vector<hipEvent_t> ev;
  double *S = nullptr;
  int S_cap = 0, S_top = 0;
  double *d_history = nullptr;
  // the solve in progress
  const storm_hip_vec *b = nullptr;
  storm_hip_vec *x = nullptr;
  int64_t n = 0;
  int inner = 0, gram_schmidt = 0, lag = 4;
  std::vector<V> work;
  const int *dp = nullptr;  // predicate of the statements being enqueued (null: unconditional)
  int status = STORM_HIP_OK;
  int64_t it_enqueued = 0, applies = 0, pre_applies = 0;
  std::vector<int64_t> applies_after, pre_after;  // totals after init ([0]) and after each enqueued iteration
  bool stepping = false, active = false;
  // pending scalar program / reduction
  SProg prog{};
  int n_imm = 0;
  bool red_pending = false;
  int red_nb = 0, red_k = 0;
  RedOut red_out{};
  // a reduction whose partials kernel is launched at flush(), with the final pass and the scalar program inside
  enum { PEND_NONE, PEND_DOTS, PEND_LIN_DOT, PEND_LIN2_DOT, PEND_VMUL_DOTS } pend = PEND_NONE;
  double *pend_z = nullptr;
  const double *pend_d = nullptr, *pend_r = nullptr;
  const double *pend_a = nullptr, *pend_w = nullptr;
  DotPtrs pend_bs{};
  LinArgs pend_lin{}, pend_lin0{};  // (pend_lin0: the held-back statement of PEND_LIN2_DOT)
  int pend_nt0 = 0;
  int pend_nt = 0, pend_yy = 0, pend_flags = 0;
  // A vector statement held back (at most one): if the next statement is one too, both go out as ONE pass
  // (lin2_kernel); it may also be overtaken by a reduction that shares no vector with it.  It is older than any
  // pending reduction / scalar program, so launching it first is always right; holding it back past a program is
  // right when the program writes none of the registers its coefficients read.
  bool q_has = false;
  LinArgs q_lin{};
  int q_nt = 0;
  int q_regs[4] = {-1, -1, -1, -1};  // registers its coefficients (and its condition) read
  long long q_gate = -1;  // >= 0: held back past the convergence rule of iteration q_gate - 1 (see lin_kernel)
  int64_t cur_it = 0;     // the iteration iterate() is enqueuing
  // Sweep directions (see storm_hip_solve_cg): every streaming statement deals its blocks out from the end of the
  // rows where the previous one stopped -- what the Infinity Cache still holds.  Same rows and slots per block.
  int sweep_dir = 1;
  int flip() { return (c->opt_sweep_alternate != 0) ? (sweep_dir ^= 1) : 0; }
  int stream_flags() { return stream_nt(c, n) | (flip() << 1); }
  // per-method vectors and registers
  V p = nullptr, q = nullptr, r = nullptr, rt = nullptr, t = nullptr, u = nullptr, v = nullptr, y = nullptr, z = nullptr,
    d = nullptr, s_ = nullptr;
  std::vector<V> qs, zs, rs, us, ps, gs;
  int r_alpha = 0, r_beta = 0, r_rho = 0, r_omega = 0, r_gamma = 0, r_tau = 0, r_a0 = 0, r_a1 = 0, r_a2 = 0, r_a3 = 0,
      r_a4 = 0;
  int H0 = 0, B0 = 0, CS0 = 0, SN0 = 0, r_hn = 0;

  bool has_pre() const { return pre_fn != nullptr || pre_diag != nullptr; }
  bool left() const { return has_pre() && side == STORM_HIP_LEFT; }
  bool right() const { return has_pre() && side == STORM_HIP_RIGHT; }
  bool ok() const { return status == STORM_HIP_OK; }
  void fail(int st) {
    if (status == STORM_HIP_OK) status = st;
  }
  int alloc(int count) {
    const int at = S_top;
    S_top += count;
    return at;
  }

  // -- scalar statements
  void sc(uint16_t opc, int dd, int aa = 0, int bb = 0) {
    if (prog.n == kProgOps) flush();
    prog.ops[prog.n++] = SOp{opc, (uint16_t)dd, (uint16_t)aa, (uint16_t)bb};
  }
  int imm(double value) {
    if (n_imm == kProgImm || prog.n + 4 > kProgOps) flush();
    prog.imm[n_imm] = value;
    return kImm0 + n_imm++;
  }
  void reset_prog() { prog.n = 0, n_imm = 0; }

  // May the held-back statement stay behind the scalar program about to go out?  0: no; 1: yes; 2: yes, and the
  // program holds this iteration's convergence rule (the statement is then gated on the iteration counter).
  int prog_lets_queued_wait() const {
    int verdict = 1;
    // (registers the pending REDUCTION writes directly count as written by the program that rides behind it)
    if (red_pending)
      for (int j = 0; j < red_k; ++j)
        for (int t = 0; t < 4; ++t)
          if (q_regs[t] == (int)red_out.idx[j]) return 0;
    for (int i = 0; i < prog.n; ++i) {
      const SOp &o = prog.ops[i];
      if (o.op == SC_GIVENS || o.op == SC_BACKSOLVE || o.op == SC_BEGIN) return 0;  // (macros over register ranges; init)
      if (o.op == SC_ADVANCE) {
        verdict = 2;
        continue;
      }
      const int span = o.op == SC_SYMORTHO ? 3 : 1;
      for (int t = 0; t < 4; ++t)
        if (q_regs[t] >= (int)o.d && q_regs[t] < (int)o.d + span) return 0;
    }
    return verdict;
  }
  void settle() {  // the held-back statement goes out alone
    if (!q_has) return;
    q_has = false;
    if (!ok()) return;
    switch (q_nt) {
      case 1: launch_lin<1, false>(q_lin, q_gate); break;
      case 2: launch_lin<2, false>(q_lin, q_gate); break;
      default: launch_lin<3, false>(q_lin, q_gate); break;
    }
    q_gate = -1;
  }
  bool queued_touches(const double *ptr, bool written) const {  // would a statement on `ptr` conflict with it?
    if (!q_has || ptr == nullptr) return false;
    if (ptr == q_lin.y) return true;
    if (written)
      for (int t = 0; t < q_nt; ++t)
        if (ptr == q_lin.v[t]) return true;
    return false;
  }
  void flush(bool keep_queued = false) {
    if (q_has) {
      const int wait = (keep_queued && ok()) ? prog_lets_queued_wait() : 0;
      if (wait == 0) settle();
      else if (wait == 2) q_gate = (long long)cur_it + 1;
    }
    if (!ok()) {
      reset_prog(), red_pending = false, q_has = false;
      return;
    }
    if (red_pending && pend != PEND_NONE) {
      const FinalPass f{c->d_tickets, c->d_ticket_sums, red_k, red_out, S, d_st, prog};
      const int nti = pend_flags;
      const dim3 g(red_nb), b(kBlock);
      if (pend == PEND_DOTS) {
#define DOTS_GO(K_) hipLaunchKernelGGL(dots_prog_kernel<K_>, g, b, 0, c->stream, n, pend_a, pend_bs, c->d_partials, dp, nti, f)
        switch (red_k) {
          case 1: DOTS_GO(1); break;
          case 2: DOTS_GO(2); break;
          case 3: DOTS_GO(3); break;
          case 4: DOTS_GO(4); break;
          case 5: DOTS_GO(5); break;
          case 6: DOTS_GO(6); break;
          case 7: DOTS_GO(7); break;
          default: DOTS_GO(8); break;
        }
#undef DOTS_GO
      } else if (pend == PEND_VMUL_DOTS) {
        hipLaunchKernelGGL(vmul_dots_prog_kernel, g, b, 0, c->stream, n, pend_z, pend_d, pend_r, c->d_partials, dp, nti, f);
      } else if (pend == PEND_LIN_DOT) {
#define LIN_GO(NT_)                                                                                                                   \
  if (pend_w != nullptr)                                                                                                              \
    hipLaunchKernelGGL((lin_dot_prog_kernel<NT_, true>), g, b, 0, c->stream, n, pend_lin, pend_w, pend_yy, c->d_partials, dp, nti, f); \
  else                                                                                                                                \
    hipLaunchKernelGGL((lin_dot_prog_kernel<NT_, false>), g, b, 0, c->stream, n, pend_lin, pend_w, pend_yy, c->d_partials, dp, nti, f)
        switch (pend_nt) {
          case 1: LIN_GO(1); break;
          case 2: LIN_GO(2); break;
          default: LIN_GO(3); break;
        }
#undef LIN_GO
      } else {
#define LIN2_GO(A_, B_)                                                                                                                                  \
  if (pend_w != nullptr)                                                                                                                                 \
    hipLaunchKernelGGL((lin2_dot_prog_kernel<A_, B_, true>), g, b, 0, c->stream, n, pend_lin0, pend_lin, pend_w, pend_yy, c->d_partials, dp, nti, f);     \
  else                                                                                                                                                   \
    hipLaunchKernelGGL((lin2_dot_prog_kernel<A_, B_, false>), g, b, 0, c->stream, n, pend_lin0, pend_lin, pend_w, pend_yy, c->d_partials, dp, nti, f)
        switch (pend_nt0 * 4 + pend_nt) {
          case 5: LIN2_GO(1, 1); break;
          case 6: LIN2_GO(1, 2); break;
          case 7: LIN2_GO(1, 3); break;
          case 9: LIN2_GO(2, 1); break;
          case 10: LIN2_GO(2, 2); break;
          case 11: LIN2_GO(2, 3); break;
          case 13: LIN2_GO(3, 1); break;
          case 14: LIN2_GO(3, 2); break;
          default: LIN2_GO(3, 3); break;
        }
#undef LIN2_GO
      }
      pend = PEND_NONE, red_pending = false;
    } else if (red_pending) {
      const double *partials = c->d_partials;
      int nb = red_nb;
      if (nb > kSinglePassPartials) {
        hipLaunchKernelGGL(reduce_stage1_kernel2, dim3(kStage2, red_k), dim3(kBlock), 0, c->stream, partials, nb,
                           c->d_partials2, dp);
        partials = c->d_partials2, nb = kStage2;
      }
      IpcDev w{};
      const bool ipc = c->comm != nullptr && comm_ipc_next(c, &w);
      if (c->comm == nullptr || ipc) {
        hipLaunchKernelGGL(reduce_prog_kernel, dim3(1), dim3(kBlock), 0, c->stream, partials, nb, red_k, red_out, S,
                           d_st, prog, dp, w, (int)ipc);
      } else {
        RedOut scr{};
        for (int j = 0; j < red_k; ++j) scr.idx[j] = R_SCR + j;
        SProg none{};
        hipLaunchKernelGGL(reduce_prog_kernel, dim3(1), dim3(kBlock), 0, c->stream, partials, nb, red_k, scr, S, d_st,
                           none, dp, w, 0);
        const int st = comm_allreduce_sum(c, S + R_SCR, red_k);
        if (st != STORM_HIP_OK) fail(st);
        hipLaunchKernelGGL(sprog_kernel, dim3(1), dim3(1), 0, c->stream, S, d_st, prog, red_k, red_out, (int)R_SCR, dp);
      }
      red_pending = false;
    } else if (prog.n > 0) {
      hipLaunchKernelGGL(sprog_kernel, dim3(1), dim3(1), 0, c->stream, S, d_st, prog, 0, RedOut{}, 0, dp);
    }
    reset_prog();
    if (hipGetLastError() != hipSuccess) {
      set_error("krylov: kernel launch failed");
      fail(STORM_HIP_E_HIP);
    }
  }

  // -- reductions: reg_j = <a, b_j>
  void dots(const storm_hip_vec *a, std::initializer_list<std::pair<int, const storm_hip_vec *>> outs) {
    std::vector<std::pair<int, const storm_hip_vec *>> o(outs);
    dots_v(a, o);
  }
  bool one_launch(int k) const { return c->comm == nullptr && c->opt_fused_reduce != 0 && n > 0 && k <= kDotChunk; }
  void dots_v(const storm_hip_vec *a, const std::vector<std::pair<int, const storm_hip_vec *>> &outs) {
    bool overtake = !queued_touches(a->d, false);  // a pure read: conflicts only with the held-back statement's target
    for (const auto &o : outs) overtake = overtake && !queued_touches(o.second->d, false);
    flush(overtake);
    if (!ok()) return;
    const int k = (int)outs.size();
    if (k < 1 || k > kMaxMulti) {
      set_error("krylov: %d simultaneous reductions (limit %d)", k, kMaxMulti);
      return fail(STORM_HIP_E_UNSUPPORTED);
    }
    const double *bs[kMaxMulti];
    for (int j = 0; j < k; ++j) bs[j] = outs[j].second->d, red_out.idx[j] = outs[j].first;
    if (one_launch(k)) {  // small: the partials kernel goes out at flush(), with the final pass in its last block
      int nb = stream_blocks(n);
      if ((int64_t)nb * k > c->partials_capacity) nb = (int)(c->partials_capacity / k);
      if ((int64_t)k * ((nb + kTicketGroup - 1) / kTicketGroup) <= (int64_t)8 * kTicketMaxGroups) {
        pend = PEND_DOTS, pend_a = a->d, pend_flags = stream_flags();
        for (int j = 0; j < kDotChunk; ++j) pend_bs.b[j] = bs[j < k ? j : 0];
        red_nb = nb, red_k = k, red_pending = true;
        return;
      }
    }
    if (n == 0) {  // an empty rank still takes part in the all-reduce
      const int st = (int)hipMemsetAsync(c->d_partials, 0, sizeof(double) * (size_t)k, c->stream);
      if (st != 0) return fail(STORM_HIP_E_HIP);
      red_nb = 1;
    } else {
      c->stream_reverse = flip();
      const int st = k_multi_dot_partials(c, a->d, bs, k, n, &red_nb, dp);
      c->stream_reverse = 0;
      if (st != STORM_HIP_OK) return fail(st);
    }
    red_k = k, red_pending = true;
  }
  void dot(int out, const storm_hip_vec *a, const storm_hip_vec *b2) { dots(a, {{out, b2}}); }

  // -- vector statements
  Scal scal(const Coef &co) const { return co.reg >= 0 ? Scal{S + co.reg, 0.0, co.sign} : Scal{nullptr, co.v, 1.0}; }
  template <int NT, bool NESTED>
  void launch_lin(const LinArgs &a, long long gate = -1) {
    if (n <= 0) return;
    const int64_t per_block = (int64_t)kBlock * lin_unroll(NT) * 2;
    const int64_t nb = std::min<int64_t>(65536, std::max<int64_t>(1, (n + per_block - 1) / per_block));
    hipLaunchKernelGGL((lin_kernel<NT, NESTED>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a, dp, stream_flags(),
                       gate >= 0 ? &d_st->iteration : nullptr, gate);
  }
  template <int NT1>
  void launch_lin2(const LinArgs &a1, const LinArgs &a2, int nt2) {
    const int64_t nb = std::min<int64_t>(131072, std::max<int64_t>(1, ((n >> 1) + kBlock - 1) / kBlock));
    const int fl = stream_flags();
    const long long *gp = q_gate >= 0 ? &d_st->iteration : nullptr;
    const long long gv = q_gate;
    switch (nt2) {
      case 1: hipLaunchKernelGGL((lin2_kernel<NT1, 1>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a1, a2, dp, fl, gp, gv); break;
      case 2: hipLaunchKernelGGL((lin2_kernel<NT1, 2>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a1, a2, dp, fl, gp, gv); break;
      default: hipLaunchKernelGGL((lin2_kernel<NT1, 3>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a1, a2, dp, fl, gp, gv); break;
    }
    q_gate = -1;
  }
  void lin_v(V yv, const std::vector<Term> &terms, int cond = -1) {
    if (c->opt_lin_fuse != 0 && terms.size() >= 1 && terms.size() <= 3 && n > 1) {
      flush(true);
      if (!ok()) return;
      LinArgs a{};
      a.y = yv->d;
      a.cond = cond >= 0 ? S + cond : nullptr;
      const int nt = (int)terms.size();
      int regs[4] = {-1, -1, -1, cond};
      for (int t = 0; t < nt; ++t) a.v[t] = terms[(size_t)t].v->d, a.c[t] = scal(terms[(size_t)t].c), regs[t] = terms[(size_t)t].c.reg;
      if (q_has) {  // the held-back statement and this one: one pass
        q_has = false;
        switch (q_nt) {
          case 1: launch_lin2<1>(q_lin, a, nt); break;
          case 2: launch_lin2<2>(q_lin, a, nt); break;
          default: launch_lin2<3>(q_lin, a, nt); break;
        }
      } else {
        q_has = true, q_lin = a, q_nt = nt;
        for (int t = 0; t < 4; ++t) q_regs[t] = regs[t];
      }
      return;
    }
    flush();
    if (!ok()) return;
    size_t at = 0;
    bool first = true;
    while (at < terms.size()) {
      LinArgs a{};
      a.y = yv->d;
      a.cond = cond >= 0 ? S + cond : nullptr;
      int nt = 0;
      if (!first) a.v[nt] = yv->d, a.c[nt] = Scal{nullptr, 1.0, 1.0}, ++nt;
      while (at < terms.size() && nt < 4) a.v[nt] = terms[at].v->d, a.c[nt] = scal(terms[at].c), ++nt, ++at;
      switch (nt) {
        case 1: launch_lin<1, false>(a); break;
        case 2: launch_lin<2, false>(a); break;
        case 3: launch_lin<3, false>(a); break;
        default: launch_lin<4, false>(a); break;
      }
      first = false;
    }
  }
  void lin(V yv, std::initializer_list<Term> terms, int cond = -1) { lin_v(yv, std::vector<Term>(terms), cond); }
  // y = v0 + c1 * (v1 + c2 * v2)
  void lin_nested(V yv, const storm_hip_vec *v0, Coef c1, const storm_hip_vec *v1, Coef c2, const storm_hip_vec *v2) {
    flush();
    if (!ok()) return;
    LinArgs a{};
    a.y = yv->d;
    a.v[0] = v0->d, a.v[1] = v1->d, a.v[2] = v2->d;
    a.c[0] = Scal{nullptr, 1.0, 1.0}, a.c[1] = scal(c1), a.c[2] = scal(c2);
    launch_lin<3, true>(a);
  }
  void copy(V yv, const storm_hip_vec *xv, int cond = -1) { lin(yv, {{num(1.0), xv}}, cond); }
  void axpy(V yv, Coef a, const storm_hip_vec *xv) { lin(yv, {{num(1.0), yv}, {a, xv}}); }
  void divide(V yv, int reg) {  // y /= reg (a true division per element, SolverGmres.hpp:88)
    flush();
    if (!ok()) return;
    c->stream_reverse = flip();
    const int st = k_scale(c, yv->d, n, dev_scal(S + reg), true, dp);
    c->stream_reverse = 0;
    if (st != STORM_HIP_OK) fail(st);
  }
  void scale(V yv, Coef a) { lin(yv, {{a, yv}}); }

  // y = sum_t c_t v_t  AND  reg_yy = <y, y>, reg_yw = <y, w> of the new y (register < 0: not wanted), one pass.
  void lin_dots(V yv, std::initializer_list<Term> terms_il, int reg_yy, int reg_yw = -1, const storm_hip_vec *wv = nullptr) {
    std::vector<Term> terms(terms_il);
    if (terms.size() > 3 || n <= 0 || (wv != nullptr && wv == yv)) {  // not this kernel's shape: two statements
      lin_v(yv, terms);
      std::vector<std::pair<int, const storm_hip_vec *>> outs;
      if (reg_yy >= 0) outs.push_back({reg_yy, yv});
      if (reg_yw >= 0) outs.push_back({reg_yw, wv});
      dots_v(yv, outs);
      return;
    }
    bool with_held = false;  // the held-back statement goes into THIS pass (it conflicts, so it cannot wait)
    {
      bool overtake = !queued_touches(yv->d, true) && !(wv != nullptr && queued_touches(wv->d, false));
      for (const Term &t : terms) overtake = overtake && !queued_touches(t.v->d, false);
      with_held = q_has && !overtake && q_gate < 0 && q_lin.cond == nullptr && c->opt_lin_fuse != 0 && one_launch(2) && n > 1 &&
                  (reg_yy >= 0 || (reg_yw >= 0 && wv != nullptr));
      if (with_held) {
        flush(true);                         // (may still settle it: a scalar program in the way)
        with_held = q_has && q_gate < 0;
        if (q_has && !with_held) settle();
      } else {
        flush(overtake);
      }
    }
    if (!ok()) return;
    LinArgs a{};
    a.y = yv->d;
    const int nt = (int)terms.size();
    for (int t = 0; t < nt; ++t) a.v[t] = terms[(size_t)t].v->d, a.c[t] = scal(terms[(size_t)t].c);
    const double *wd = (reg_yw >= 0 && wv != nullptr) ? wv->d : nullptr;
    const int64_t per_block = (int64_t)kBlock * lin_unroll(nt + (wd != nullptr ? 1 : 0)) * 2;  // (the kernels' U: HASW)
    int64_t nb = std::max<int64_t>(1, (n + per_block - 1) / per_block);
    nb = std::min<int64_t>(nb, std::min<int64_t>(32768, c->partials_capacity / 2));
    const int nti = stream_flags();
    if (with_held) {
      const int64_t nb2 = nb;  // the grid of the statement alone (same rows per block, same partial sums)
      pend = PEND_LIN2_DOT, pend_lin0 = q_lin, pend_nt0 = q_nt, q_has = false;
      pend_lin = a, pend_nt = nt, pend_w = wd, pend_yy = (int)(reg_yy >= 0), pend_flags = nti;
      red_k = 0;
      if (reg_yy >= 0) red_out.idx[red_k++] = reg_yy;
      if (wd != nullptr) red_out.idx[red_k++] = reg_yw;
      red_nb = (int)nb2, red_pending = true;
      return;
    }
    if (one_launch(2) && (reg_yy >= 0 || wd != nullptr)) {
      pend = PEND_LIN_DOT, pend_lin = a, pend_nt = nt, pend_w = wd, pend_yy = (int)(reg_yy >= 0), pend_flags = nti;
      red_k = 0;
      if (reg_yy >= 0) red_out.idx[red_k++] = reg_yy;
      if (wd != nullptr) red_out.idx[red_k++] = reg_yw;
      red_nb = (int)nb, red_pending = true;
      return;
    }
    switch (nt) {
      case 1: if (wd != nullptr) hipLaunchKernelGGL((lin_dot_kernel<1, true>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a, wd, (int)(reg_yy >= 0), c->d_partials, dp, nti);
        else hipLaunchKernelGGL((lin_dot_kernel<1, false>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a, wd, (int)(reg_yy >= 0), c->d_partials, dp, nti);
        break;
      case 2: if (wd != nullptr) hipLaunchKernelGGL((lin_dot_kernel<2, true>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a, wd, (int)(reg_yy >= 0), c->d_partials, dp, nti);
        else hipLaunchKernelGGL((lin_dot_kernel<2, false>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a, wd, (int)(reg_yy >= 0), c->d_partials, dp, nti);
        break;
      default: if (wd != nullptr) hipLaunchKernelGGL((lin_dot_kernel<3, true>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a, wd, (int)(reg_yy >= 0), c->d_partials, dp, nti);
        else hipLaunchKernelGGL((lin_dot_kernel<3, false>), dim3((int)nb), dim3(kBlock), 0, c->stream, n, a, wd, (int)(reg_yy >= 0), c->d_partials, dp, nti);
        break;
    }
    red_k = 0;
    if (reg_yy >= 0) red_out.idx[red_k++] = reg_yy;
    if (wd != nullptr) red_out.idx[red_k++] = reg_yw;
    red_nb = (int)nb, red_pending = red_k > 0;
  }
  // y = A(x)  AND  reg_wy = <w, y>, reg_yy = <y, y> (register < 0: not wanted): the stencil SpMV's fused epilogue
  // when the operator is native and has no CSR tail, separate reductions otherwise.
  void apply_dots(V yv, const storm_hip_vec *xv, int reg_wy, const storm_hip_vec *wv, int reg_yy = -1) {
    const bool fusable = op_fn == nullptr && op != nullptr && op->tail_rows == 0 && c->opt_fuse_dot != 0 && n > 0 &&
                         reg_wy >= 0;
    if (!fusable) {
      apply(yv, xv);
      std::vector<std::pair<int, const storm_hip_vec *>> outs;
      if (reg_wy >= 0) outs.push_back({reg_wy, wv});
      if (reg_yy >= 0) outs.push_back({reg_yy, yv});
      dots_v(yv, outs);
      return;
    }
    flush();
    if (!ok()) return;
    ++applies;
    int nblocks = 0;
    SpmvDot sd;
    sd.w = wv->d, sd.yy = reg_yy >= 0, sd.partials = c->d_partials, sd.nblocks_out = &nblocks;
    c->spmv_reverse = flip();
    const int st = spmv_launch(op, host_scal(op_alpha), host_scal(op_beta), xv->d, yv->d, &sd, dp);
    c->spmv_reverse = 0;
    if (st != STORM_HIP_OK) return fail(st);
    if (nblocks <= 0) {  // the launch did not fuse after all
      std::vector<std::pair<int, const storm_hip_vec *>> outs{{reg_wy, wv}};
      if (reg_yy >= 0) outs.push_back({reg_yy, yv});
      dots_v(yv, outs);
      return;
    }
    red_k = 0;
    red_out.idx[red_k++] = reg_wy;
    if (reg_yy >= 0) red_out.idx[red_k++] = reg_yy;
    red_nb = nblocks, red_pending = true;
  }

  struct ApiDone {  // library calls a callback makes are predicated on this solve's flag
    storm_hip_ctx *c;
    const int *saved;
    ApiDone(storm_hip_ctx *c_, const int *dp_) : c(c_), saved(c_->api_done) { c->api_done = dp_, ++c->callback_depth; }
    ~ApiDone() { c->api_done = saved, --c->callback_depth; }
  };
  void apply(V yv, const storm_hip_vec *xv) {  // y = A(x)          Operator::mul, Operator.hpp:74
    flush();
    if (!ok()) return;
    ++applies;
    int st;
    if (op_fn != nullptr) {
      ApiDone guard(c, dp);
      st = op_fn(op_user, yv, xv);
      if (st != 0) {
        if (st > 0 || storm_hip_last_error()[0] == 0) set_error("krylov: the operator callback returned %d", st);
        st = st < 0 ? st : STORM_HIP_E_INVALID;
      }
    } else {
      c->spmv_reverse = flip();
      st = spmv_launch(op, host_scal(op_alpha), host_scal(op_beta), xv->d, yv->d, nullptr, dp);
      c->spmv_reverse = 0;
    }
    if (st != STORM_HIP_OK) fail(st);
  }
  void pre(V yv, const storm_hip_vec *xv) {  // y = P(x)          Preconditioner::mul
    flush();
    if (!ok()) return;
    ++pre_applies;
    int st;
    ApiDone guard(c, dp);
    if (pre_fn != nullptr) {
      st = pre_fn(pre_user, yv, xv);
      if (st != 0) {
        if (st > 0 || storm_hip_last_error()[0] == 0) set_error("krylov: the preconditioner callback returned %d", st);
        st = st < 0 ? st : STORM_HIP_E_INVALID;
      }
    } else {
      c->stream_reverse = flip();
      st = storm_hip_vmul(yv, pre_diag, xv);
      c->stream_reverse = 0;
    }
    if (st != STORM_HIP_OK) fail(st);
  }
  // z = P(r) AND reg_rz = <r, z>, reg_rr = <r, r>: one pass when the preconditioner is the library's diagonal one.
  void pre_dots(V zv, const storm_hip_vec *rv, int reg_rz, int reg_rr) {
    if (pre_diag == nullptr || c->opt_lin_fuse == 0 || !one_launch(2) || zv == rv) {
      pre(zv, rv);
      dots(rv, {{reg_rz, zv}, {reg_rr, rv}});
      return;
    }
    flush();
    if (!ok()) return;
    ++pre_applies;
    int nb = stream_blocks(n);
    if ((int64_t)nb * 2 > c->partials_capacity) nb = (int)(c->partials_capacity / 2);
    pend = PEND_VMUL_DOTS, pend_z = zv->d, pend_d = pre_diag->d, pend_r = rv->d, pend_flags = stream_flags();
    red_k = 2, red_out.idx[0] = reg_rz, red_out.idx[1] = reg_rr;
    red_nb = nb, red_pending = true;
  }
  // The dispatch every preconditioned body repeats (chained mul, Operator.hpp:82-88):
  //   left: z = P(y = A x);  right: z = A(y = P x);  none: z = A x.
  void mul_side(V zv, V yv, const storm_hip_vec *xv) {
    if (left()) apply(yv, xv), pre(zv, yv);
    else if (right()) pre(yv, xv), apply(zv, yv);
    else apply(zv, xv);
  }
  void residual(V rv, const storm_hip_vec *bv, const storm_hip_vec *xv) {  // Operator::Residual, Operator.hpp:95-99
    apply(rv, xv);
    lin(rv, {{num(1.0), bv}, {num(-1.0), rv}});
  }
  V vec() {
    storm_hip_vec *w = nullptr;
    if (ok()) {
      const int st = storm_hip_vec_create_like(x, &w);
      if (st != STORM_HIP_OK) fail(st);
    }
    work.push_back(w);
    return w;
  }
  void norm_to_err_and(uint16_t what, const storm_hip_vec *a) {  // ERR = |a|; begin / advance
    dot(R_T0, a, a);
    sc(SC_SQRT, R_ERR, R_T0);
    sc(what, 0, R_ERR);
  }

  // ---- the solvers -------------------------------------------------------------------------------------------
  void setup();
  void init();
  void iterate(int64_t it);
  void finalize(int64_t iterations, bool forced);
  void gmres_start(bool outer);
  void gmres_update_x(int k);
};

struct storm_hip_krylov : KrylovEngine {};  // the opaque handle of the C ABI

namespace {
typedef KrylovEngine K;
}

// Registers and work vectors of a method (the reference allocates in init(): e.g. SolverCg.hpp:57-59).
void K::setup() {
  S_top = R_USER;
  const bool P = has_pre();
  switch (method) {
    case STORM_HIP_CG:
      p = vec(), r = vec(), z = vec();
      r_gamma = alloc(1), r_alpha = alloc(1), r_beta = alloc(1), r_a0 = alloc(1), r_a1 = alloc(1);
      break;
    case STORM_HIP_BICGSTAB:
      p = vec(), r = vec(), rt = vec(), t = vec(), v = vec();
      if (P) z = vec();
      r_alpha = alloc(1), r_beta = alloc(1), r_rho = alloc(1), r_omega = alloc(1), r_a0 = alloc(1), r_a1 = alloc(1),
      r_a2 = alloc(1);
      break;
    case STORM_HIP_CGS:
      p = vec(), q = vec(), r = vec(), rt = vec(), u = vec(), v = vec();
      r_alpha = alloc(1), r_beta = alloc(1), r_rho = alloc(1), r_a0 = alloc(1), r_a1 = alloc(1);
      break;
    case STORM_HIP_TFQMR:
    case STORM_HIP_TFQMR1:
      d = vec(), rt = vec(), u = vec(), v = vec(), y = vec(), s_ = vec();
      if (P) z = vec();
      r_alpha = alloc(1), r_beta = alloc(1), r_rho = alloc(1), r_tau = alloc(1), r_omega = alloc(1), r_a0 = alloc(1),
      r_a1 = alloc(3), r_a2 = alloc(1), r_a3 = alloc(1), r_a4 = alloc(1);
      break;
    case STORM_HIP_RICHARDSON:
      r = vec();
      if (P) z = vec();
      break;
    case STORM_HIP_BICGSTAB_L: {
      const int l = inner;
      rt = vec();
      if (P) z = vec();
      rs.clear(), us.clear();
      for (int i = 0; i <= l; ++i) rs.push_back(vec()), us.push_back(vec());
      r_alpha = alloc(1), r_beta = alloc(1), r_rho = alloc(1), r_omega = alloc(1), r_a0 = alloc(1);
      r_gamma = alloc(l + 1);           // gamma
      r_a1 = alloc(l + 1);              // gamma_bar
      r_a2 = alloc(l + 1);              // gamma_bbar
      r_a3 = alloc(l + 1);              // sigma
      r_a4 = alloc((l + 1) * (l + 1));  // tau
    } break;
    case STORM_HIP_IDRS: {
      const int s = inner;
      r = vec(), v = vec();
      if (P) z = vec();
      ps.clear(), us.clear(), gs.clear();
      for (int i = 0; i < s; ++i) ps.push_back(vec()), us.push_back(vec()), gs.push_back(vec());
      r_omega = alloc(1), r_alpha = alloc(1), r_beta = alloc(1);
      r_a0 = alloc(s);      // phi
      r_gamma = alloc(s);   // gamma
      r_a1 = alloc(s * s);  // mu
    } break;
    case STORM_HIP_GMRES:
    case STORM_HIP_FGMRES: {
      const int m = inner;
      qs.clear(), zs.clear();
      for (int i = 0; i <= m; ++i) qs.push_back(vec());
      if (P) {
        const int nz = method == STORM_HIP_FGMRES ? m : 1;
        for (int i = 0; i < nz; ++i) zs.push_back(vec());
      }
      r_hn = alloc(1);
      B0 = alloc(m + 1), CS0 = alloc(m), SN0 = alloc(m);
      r_a0 = alloc(2 * kMaxMulti);  // classical Gram-Schmidt x2: the two passes' coefficients
      H0 = alloc((m + 1) * m);
    } break;
    default:
      set_error("krylov: unknown method %d", method);
      fail(STORM_HIP_E_INVALID);
  }
}

// ---- GMRES / FGMRES pieces ----------------------------------------------------------------------------------
// q0 = b - A x [left: q0 = P(b - A x)]; beta0 = |q0|; q0 /= beta0      (outer_init :82-90 and inner_init :110-116)
void K::gmres_start(bool outer) {
  const bool lp = has_pre() && method == STORM_HIP_GMRES && side == STORM_HIP_LEFT;
  residual(qs[0], b, x);
  if (lp) {
    std::swap(zs[0], qs[0]);
    pre(qs[0], zs[0]);
  }
  dot(R_T0, qs[0], qs[0]);
  sc(SC_SQRT, B0, R_T0);
  if (outer) sc(SC_BEGIN, 0, B0);
  divide(qs[0], B0);
}

// x += sum_i beta_i q_i after the back substitution (inner_finalize :194-249)
void K::gmres_update_x(int k) {
  const bool rp = has_pre() && (method == STORM_HIP_FGMRES || side == STORM_HIP_RIGHT);
  prog.aux[0] = H0, prog.aux[1] = B0, prog.aux[2] = CS0, prog.aux[3] = SN0, prog.aux[4] = inner;
  sc(SC_BACKSOLVE, 0, k);
  std::vector<Term> terms;
  if (!rp) {
    for (int i = 0; i <= k; ++i) terms.push_back({R(B0 + i), qs[i]});
    terms.insert(terms.begin(), Term{num(1.0), x});
    lin_v(x, terms);
  } else if (method == STORM_HIP_FGMRES) {
    terms.push_back({num(1.0), x});
    for (int i = 0; i <= k; ++i) terms.push_back({R(B0 + i), zs[i]});
    lin_v(x, terms);
  } else {  // q0 = sum beta_i q_i; z0 = P q0; x += z0          :242-247
    for (int i = 0; i <= k; ++i) terms.push_back({R(B0 + i), qs[i]});
    lin_v(qs[0], terms);
    pre(zs[0], qs[0]);
    axpy(x, num(1.0), zs[0]);
  }
}

namespace storm {
// solvers.hip: the Gram-Schmidt step of storm_hip_solve_gmres, shared with this engine
int gmres_orthogonalize(storm_hip_ctx *c, int64_t n, const SolverState *st, const int *done, double *qn,
                        const double *const *q, int k, int m, double *H, double *norm2_out, double *scratch,
                        int gram_schmidt, bool *normalised, const MgsGivens *givens = nullptr,
                        bool *givens_done = nullptr, const ChainApply *apply = nullptr);
}  // namespace storm

void K::init() {
  const bool P = has_pre();
  switch (method) {
    case STORM_HIP_CG:  // SolverCg.hpp:54-84 (CG takes no notice of pre_side)
      residual(r, b, x);
      if (P) {
        pre(z, r);
        copy(p, z);
        dots(r, {{r_gamma, z}, {R_T0, r}});
        sc(SC_SQRT, R_ERR, R_T0);
      } else {
        copy(p, r);
        dot(r_gamma, r, r);
        sc(SC_SQRT, R_ERR, r_gamma);
      }
      sc(SC_BEGIN, 0, R_ERR);
      break;
    case STORM_HIP_BICGSTAB:  // SolverBiCgStab.hpp:59-91
      residual(r, b, x);
      if (left()) std::swap(z, r), pre(r, z);
      copy(rt, r);
      dot(r_rho, rt, r);
      sc(SC_SQRT, R_ERR, r_rho);
      sc(SC_BEGIN, 0, R_ERR);
      break;
    case STORM_HIP_CGS:  // SolverCgs.hpp:54-88
      residual(r, b, x);
      if (left()) std::swap(u, r), pre(r, u);
      copy(rt, r);
      dot(r_rho, rt, r);
      sc(SC_SQRT, R_ERR, r_rho);
      sc(SC_BEGIN, 0, R_ERR);
      break;
    case STORM_HIP_TFQMR:
    case STORM_HIP_TFQMR1:  // SolverTfqmr.hpp:41-87
      if (method == STORM_HIP_TFQMR1) copy(d, x);
      residual(y, b, x);
      if (left()) std::swap(z, y), pre(y, z);
      copy(u, y);
      copy(rt, u);
      dot(r_rho, rt, u);
      sc(SC_SQRT, r_tau, r_rho);
      sc(SC_BEGIN, 0, r_tau);
      break;
    case STORM_HIP_RICHARDSON:  // SolverRichardson.hpp:48-71 (no notice of pre_side either)
      residual(r, b, x);
      if (P) std::swap(z, r), pre(r, z);
      norm_to_err_and(SC_BEGIN, r);
      break;
    case STORM_HIP_BICGSTAB_L:  // SolverBiCgStab.hpp:195-233 (always left)
      residual(rs[0], b, x);
      if (P) std::swap(z, rs[0]), pre(rs[0], z);
      copy(rt, rs[0]);
      dot(r_rho, rt, rs[0]);
      sc(SC_SQRT, R_ERR, r_rho);
      sc(SC_BEGIN, 0, R_ERR);
      break;
    case STORM_HIP_IDRS:  // SolverIdrs.hpp:60-106
      residual(r, b, x);
      if (left()) std::swap(z, r), pre(r, z);
      dot(R_T0, r, r);
      sc(SC_SQRT, r_a0, R_T0);
      sc(SC_BEGIN, 0, r_a0);
      break;
    case STORM_HIP_GMRES:
    case STORM_HIP_FGMRES:  // SolverGmres.hpp:51-91
      gmres_start(true);
      break;
  }
  flush();
}

void K::iterate(int64_t it) {
  cur_it = it;
  const bool P = has_pre();
  switch (method) {
    case STORM_HIP_CG: {  // SolverCg.hpp:86-126
      apply_dots(z, p, R_T0, p);
      sc(SC_SDIV, r_alpha, r_gamma, R_T0);
      sc(SC_MOV, r_a0, r_gamma);  // gamma_bar
      axpy(x, R(r_alpha), p);
      if (P) {
        axpy(r, mR(r_alpha), z);
        pre_dots(z, r, r_gamma, R_T1);
        sc(SC_SQRT, R_ERR, R_T1);
      } else {
        lin_dots(r, {{num(1.0), r}, {mR(r_alpha), z}}, r_gamma);
        sc(SC_SQRT, R_ERR, r_gamma);
      }
      sc(SC_SDIV, r_beta, r_gamma, r_a0);
      sc(SC_ADVANCE, 0, R_ERR);
      lin(p, {{num(1.0), P ? z : r}, {R(r_beta), p}});
    } break;

    case STORM_HIP_BICGSTAB: {  // SolverBiCgStab.hpp:93-165
      if (it == 0) {
        copy(p, r);
      } else {
        lin_nested(p, r, R(r_beta), p, mR(r_omega), v);  // (rho, beta: formed at the end of the previous iteration)
      }
      if (left()) {
        mul_side(v, z, p);
        dot(R_T0, rt, v);
      } else {  // the operator is applied last: its reduction rides in the SpMV
        if (right()) pre(z, p);
        apply_dots(v, right() ? z : p, R_T0, rt);
      }
      sc(SC_SDIV, r_alpha, r_rho, R_T0);
      // (:140, :161: without a preconditioner p and s = r are still there when omega is known, and
      //  x = (x + alpha p) + omega s goes out as ONE statement below -- the same two roundings per element)
      if (P) axpy(x, R(r_alpha), right() ? z : p);
      axpy(r, mR(r_alpha), v);
      if (left()) {
        mul_side(t, z, r);
        dots(t, {{R_T0, r}, {R_T1, t}});
      } else {
        if (right()) pre(z, r);
        apply_dots(t, right() ? z : r, R_T0, r, R_T1);
      }
      sc(SC_SDIV, r_omega, R_T0, R_T1);
      if (P) axpy(x, R(r_omega), right() ? z : r);
      else lin(x, {{num(1.0), x}, {R(r_alpha), p}, {R(r_omega), r}});
      lin_dots(r, {{num(1.0), r}, {mR(r_omega), t}}, R_T0, r_a1, rt);  // |r|^2 and the next iteration's <rt, r>
      sc(SC_SQRT, R_ERR, R_T0);
      sc(SC_ADVANCE, 0, R_ERR);
      // :116-118 of the NEXT iteration (the same r): rho_bar = rho; rho = <rt, r>; beta = (alpha rho) / (omega rho_bar)
      // -- in this pass's scalar program instead of a launch of their own at the start of the next iteration
      sc(SC_MOV, r_a0, r_rho);
      sc(SC_MOV, r_rho, r_a1);
      sc(SC_MUL, R_T0, r_alpha, r_rho);
      sc(SC_MUL, R_T1, r_omega, r_a0);
      sc(SC_SDIV, r_beta, R_T0, R_T1);
    } break;

    case STORM_HIP_CGS: {  // SolverCgs.hpp:90-172
      if (it == 0) {
        copy(u, r);
        copy(p, u);
      } else {
        lin(u, {{num(1.0), r}, {R(r_beta), q}});  // (rho, beta: formed at the end of the previous iteration)
        lin_nested(p, u, R(r_beta), q, R(r_beta), p);
      }
      mul_side(v, q, p);
      dot(R_T0, rt, v);
      sc(SC_SDIV, r_alpha, r_rho, R_T0);
      lin(q, {{num(1.0), u}, {mR(r_alpha), v}});
      lin(v, {{num(1.0), u}, {num(1.0), q}});
      const storm_hip_vec *step = v;  // what r loses alpha times of
      if (left()) {
        axpy(x, R(r_alpha), v);
        apply(u, v), pre(v, u);
      } else if (right()) {
        pre(u, v), apply(v, u);
        axpy(x, R(r_alpha), u);
      } else {
        apply(u, v);
        axpy(x, R(r_alpha), v);
        step = u;
      }
      lin_dots(r, {{num(1.0), r}, {mR(r_alpha), step}}, R_T0, r_a1, rt);  // |r|^2 and the next iteration's <rt, r>
      sc(SC_SQRT, R_ERR, R_T0);
      sc(SC_ADVANCE, 0, R_ERR);
      sc(SC_MOV, r_a0, r_rho);  // SolverCgs.hpp:116-118 of the next iteration, in this pass's scalar program
      sc(SC_MOV, r_rho, r_a1);
      sc(SC_SDIV, r_beta, r_rho, r_a0);
    } break;

    case STORM_HIP_TFQMR:
    case STORM_HIP_TFQMR1: {  // SolverTfqmr.hpp:89-204
      const bool l1 = method == STORM_HIP_TFQMR1;
      if (it == 0) {
        mul_side(s_, z, y);
        copy(v, s_);
      } else {
        sc(SC_MOV, r_a0, r_rho);
        sc(SC_MOV, r_rho, r_a4);  // <rt, u>: formed in the pass that produced this u (second half-step below)
        sc(SC_SDIV, r_beta, r_rho, r_a0);
        lin(v, {{num(1.0), s_}, {R(r_beta), v}});
        lin(y, {{num(1.0), u}, {R(r_beta), y}});
        mul_side(s_, z, y);
        lin(v, {{num(1.0), s_}, {R(r_beta), v}});
      }
      dot(R_T0, rt, v);
      sc(SC_SDIV, r_alpha, r_rho, R_T0);
      for (int half = 0; half <= 1; ++half) {
        axpy(d, R(r_alpha), right() ? z : y);
        if (half == 1) lin_dots(u, {{num(1.0), u}, {mR(r_alpha), s_}}, R_T0, r_a4, rt);  // + the next <rt, u>
        else lin_dots(u, {{num(1.0), u}, {mR(r_alpha), s_}}, R_T0);
        sc(SC_SQRT, r_omega, R_T0);
        if (l1) {
          sc(SC_LT, r_a2, r_omega, r_tau);
          sc(SC_CMOV, r_tau, r_omega, r_a2);
          copy(x, d, r_a2);
        } else {
          sc(SC_SYMORTHO, r_a1, r_tau, r_omega);  // (cs, sn, rr) in r_a1 .. r_a1 + 2
          sc(SC_MUL, r_tau, r_omega, r_a1);
          sc(SC_MUL, r_a2, r_a1, r_a1);            // cs^2
          sc(SC_MUL, r_a3, r_a1 + 1, r_a1 + 1);    // sn^2
          axpy(x, R(r_a2), d);
          scale(d, R(r_a3));
        }
        if (half == 0) {
          axpy(y, mR(r_alpha), v);
          mul_side(s_, z, y);
        }
      }
      if (l1) {
        sc(SC_ADVANCE, 0, r_tau);
      } else {
        sc(SC_MUL, R_ERR, r_tau, imm(std::sqrt(2.0 * (double)it + 3.0)));
        sc(SC_ADVANCE, 0, R_ERR);
      }
    } break;

    case STORM_HIP_RICHARDSON: {  // SolverRichardson.hpp:73-96
      axpy(x, num(relaxation), r);
      residual(r, b, x);
      if (P) std::swap(z, r), pre(r, z);
      norm_to_err_and(SC_ADVANCE, r);
    } break;

    case STORM_HIP_BICGSTAB_L: {  // SolverBiCgStab.hpp:235-367
      const int l = inner, j = (int)(it % l);
      const int G = r_gamma, GB = r_a1, GBB = r_a2, SG = r_a3;
      auto TAU = [&](int i, int jj) { return r_a4 + i * (l + 1) + jj; };
      if (it == 0) {
        copy(us[0], rs[0]);
      } else {
        sc(SC_MOV, r_a0, r_rho);
        dot(r_rho, rt, rs[j]);
        sc(SC_MUL, R_T0, r_alpha, r_rho);
        sc(SC_SDIV, r_beta, R_T0, r_a0);
        for (int i = 0; i <= j; ++i) lin(us[i], {{num(1.0), rs[i]}, {mR(r_beta), us[i]}});
      }
      if (P) apply(z, us[j]), pre(us[j + 1], z);
      else apply(us[j + 1], us[j]);
      dot(R_T0, rt, us[j + 1]);
      sc(SC_SDIV, r_alpha, r_rho, R_T0);
      for (int i = 0; i <= j; ++i) axpy(rs[i], mR(r_alpha), us[i + 1]);
      axpy(x, R(r_alpha), us[0]);
      if (P) apply(z, rs[j]), pre(rs[j + 1], z);
      else apply(rs[j + 1], rs[j]);
      if (j == l - 1) {
        for (int jj = 1; jj <= l; ++jj) {
          for (int i = 1; i < jj; ++i) {
            dot(R_T0, rs[i], rs[jj]);
            sc(SC_SDIV, TAU(i, jj), R_T0, SG + i);
            axpy(rs[jj], mR(TAU(i, jj)), rs[i]);
          }
          dots(rs[jj], {{SG + jj, rs[jj]}, {R_T0, rs[0]}});
          sc(SC_SDIV, GB + jj, R_T0, SG + jj);
        }
        sc(SC_MOV, G + l, GB + l);
        sc(SC_MOV, r_omega, G + l);
        sc(SC_NEG, R_T0, r_omega);
        sc(SC_MUL, r_rho, r_rho, R_T0);
        for (int jj = l - 1; jj != 0; --jj) {
          sc(SC_MOV, G + jj, GB + jj);
          for (int i = jj + 1; i <= l; ++i) sc(SC_FMSUB, G + jj, TAU(jj, i), G + i);
        }
        for (int jj = 1; jj < l; ++jj) {
          sc(SC_MOV, GBB + jj, G + jj + 1);
          for (int i = jj + 1; i < l; ++i) sc(SC_FMADD, GBB + jj, TAU(jj, i), G + i + 1);
        }
        axpy(x, R(G + 1), rs[0]);
        axpy(rs[0], mR(GB + l), rs[l]);
        axpy(us[0], mR(G + l), us[l]);
        for (int jj = 1; jj < l; ++jj) {
          axpy(x, R(GBB + jj), rs[jj]);
          axpy(rs[0], mR(GB + jj), rs[jj]);
          axpy(us[0], mR(G + jj), us[jj]);
        }
      }
      norm_to_err_and(SC_ADVANCE, rs[0]);
    } break;

    case STORM_HIP_IDRS: {  // SolverIdrs.hpp:109-281
      const int s = inner, k = (int)(it % s);
      const int PHI = r_a0, GAM = r_gamma;
      auto MU = [&](int i, int jj) { return r_a1 + i * s + jj; };
      if (k == 0) {  // inner_init :109-156
        if (it == 0) {
          sc(SC_MOV, r_omega, R_ONE);
          sc(SC_MOV, MU(0, 0), R_ONE);
          copy(ps[0], r);
          divide(ps[0], PHI);
          for (int i = 1; i < s; ++i) {
            sc(SC_MOV, MU(i, i), R_ONE);
            sc(SC_MOV, PHI + i, R_ZERO);
            flush();
            if (ok()) {
              const int st = storm_hip_fill_randomly(ps[i]);
              if (st != STORM_HIP_OK) fail(st);
            }
            for (int jj = 0; jj < i; ++jj) {
              sc(SC_MOV, MU(i, jj), R_ZERO);
              dot(R_T0, ps[i], ps[jj]);
              axpy(ps[i], mR(R_T0), ps[jj]);
            }
            dot(R_T0, ps[i], ps[i]);
            sc(SC_SQRT, R_T1, R_T0);
            divide(ps[i], R_T1);
          }
        } else {
          std::vector<std::pair<int, const storm_hip_vec *>> outs;
          for (int i = 0; i < s; ++i) outs.push_back({PHI + i, ps[i]});
          dots_v(r, outs);
        }
      }
      for (int i = k; i < s; ++i) {  // :182-188
        sc(SC_MOV, GAM + i, PHI + i);
        for (int jj = k; jj < i; ++jj) sc(SC_FMSUB, GAM + i, MU(i, jj), GAM + jj);
        sc(SC_DIV, GAM + i, GAM + i, MU(i, i));
      }
      {
        std::vector<Term> tv{{num(1.0), r}};
        for (int i = k; i < s; ++i) tv.push_back({mR(GAM + i), gs[i]});
        lin_v(v, tv);  // :200-203
      }
      if (right()) std::swap(z, v), pre(v, z);
      {
        std::vector<Term> tu{{R(r_omega), v}, {R(GAM + k), us[k]}};
        for (int i = k + 1; i < s; ++i) tu.push_back({R(GAM + i), us[i]});
        lin_v(us[k], tu);  // :208-211
      }
      if (left()) apply(z, us[k]), pre(gs[k], z);
      else apply(gs[k], us[k]);
      for (int i = 0; i < k; ++i) {  // :230-235
        dot(R_T0, ps[i], gs[k]);
        sc(SC_SDIV, r_alpha, R_T0, MU(i, i));
        axpy(us[k], mR(r_alpha), us[i]);
        axpy(gs[k], mR(r_alpha), gs[i]);
      }
      {
        std::vector<std::pair<int, const storm_hip_vec *>> outs;
        for (int i = k; i < s; ++i) outs.push_back({MU(i, k), ps[i]});
        dots_v(gs[k], outs);  // :236-238
      }
      sc(SC_SDIV, r_beta, PHI + k, MU(k, k));
      for (int i = k + 1; i < s; ++i) sc(SC_FMSUB, PHI + i, r_beta, MU(i, k));
      axpy(x, R(r_beta), us[k]);
      axpy(r, mR(r_beta), gs[k]);
      if (k == s - 1) {  // :256-279
        mul_side(v, z, r);
        dots(v, {{R_T0, r}, {R_T1, v}});
        sc(SC_SDIV, r_omega, R_T0, R_T1);
        axpy(x, R(r_omega), right() ? z : r);
        axpy(r, mR(r_omega), v);
      }
      norm_to_err_and(SC_ADVANCE, r);
    } break;

    case STORM_HIP_GMRES:
    case STORM_HIP_FGMRES: {  // Solver.hpp:236-248 around SolverGmres.hpp:119-192
      const int m = inner, k = (int)(it % m);
      const bool flexible = method == STORM_HIP_FGMRES;
      const bool lp = P && !flexible && side == STORM_HIP_LEFT, rp = P && (flexible || side == STORM_HIP_RIGHT);
      if (k == 0) gmres_start(false);
      V qn = qs[k + 1];
      if (lp) apply(zs[0], qs[k]), pre(qn, zs[0]);
      else if (rp) pre(zs[flexible ? k : 0], qs[k]), apply(qn, zs[flexible ? k : 0]);
      else apply(qn, qs[k]);
      flush();
      bool normalised = false;
      if (ok()) {
        std::vector<const double *> qd(m + 1);
        for (int i = 0; i <= m; ++i) qd[i] = qs[i]->d;
        const int st = gmres_orthogonalize(c, n, d_st, dp, qn->d, qd.data(), k, m, S + H0, S + R_T0, S + r_a0,
                                           gram_schmidt, &normalised);
        if (st != STORM_HIP_OK) fail(st);
      }
      sc(SC_SQRT, r_hn, R_T0);
      if (!normalised) divide(qn, r_hn);
      prog.aux[0] = H0, prog.aux[1] = B0, prog.aux[2] = CS0, prog.aux[3] = SN0, prog.aux[4] = m;
      sc(SC_GIVENS, R_ERR, k, r_hn);
      sc(SC_ADVANCE, 0, R_ERR);
      if (k == m - 1) {
        flush();
        gmres_update_x(k);
      }
    } break;
  }
  flush();
}

// IterativeSolver::finalize (none of the plain solvers has one) / InnerOuterIterativeSolver::finalize, Solver.hpp:250-257.
void K::finalize(int64_t iterations, bool forced) {
  if (method != STORM_HIP_GMRES && method != STORM_HIP_FGMRES) return;
  // The in-loop update of the last iteration was skipped by the `done` predicate (or, when stepping, must not be
  // repeated: it already ran if that iteration closed a restart cycle).  With no iterate() at all the reference
  // divides by H(0,0) = 0 here; not reproduced.
  if (iterations <= 0) return;
  const int k = (int)((iterations - 1) % inner);
  if (!forced && k == inner - 1) return;
  const int *saved = dp;
  dp = nullptr;
  gmres_update_x(k);
  flush();
  dp = saved;
}

// ---- host-side driving -----------------------------------------------------------------------------------------
namespace {

int check_ready(K *k, const storm_hip_vec *b, storm_hip_vec *x, const storm_hip_solver_params *p) {
  STORM_REQUIRE(k && b && x && p, "krylov: null argument");
  STORM_REQUIRE(k->op != nullptr || k->op_fn != nullptr, "krylov: no operator set");
  STORM_REQUIRE(b->ctx == k->c && x->ctx == k->c, "krylov: context mismatch");
  STORM_REQUIRE(b != x && b->d != x->d, "krylov: b and x must not alias");
  STORM_REQUIRE(b->n_owned == x->n_owned, "krylov: b has %lld rows, x %lld", (long long)b->n_owned,
                (long long)x->n_owned);
  if (k->op) {
    STORM_REQUIRE(k->op->ctx == k->c, "krylov: operator belongs to another context");
    STORM_REQUIRE(x->n_owned == k->op->n_rows, "krylov: operator has %lld rows, x %lld", (long long)k->op->n_rows,
                  (long long)x->n_owned);
    STORM_REQUIRE(x->n_halo >= k->op->n_halo, "krylov: x has %lld halo rows, operator needs %lld",
                  (long long)x->n_halo, (long long)k->op->n_halo);
  }
  if (k->pre_diag) STORM_REQUIRE(k->pre_diag->n_owned == x->n_owned, "krylov: diagonal preconditioner size mismatch");
  STORM_REQUIRE(p->num_iterations >= 0, "krylov: num_iterations < 0");
  return STORM_HIP_OK;
}

void release_work(K *k) {
  for (auto *w : k->work) storm_hip_vec_destroy(w);
  k->work.clear();
  k->qs.clear(), k->zs.clear(), k->rs.clear(), k->us.clear(), k->ps.clear(), k->gs.clear();
  if (k->d_history) (void)hipFree(k->d_history), k->d_history = nullptr;
  k->active = false;
}

// Common start of solve() and init(): state, registers, work vectors, init() enqueued.
int begin_solve(K *k, const storm_hip_vec *b, storm_hip_vec *x, const storm_hip_solver_params *p, bool stepping,
                double *history) {
  storm_hip_ctx *c = k->c;
  STORM_TRY(check_ready(k, b, x, p));
  HIP_TRY(hipSetDevice(c->device));
  release_work(k);
  k->b = b, k->x = x, k->n = x->n_owned, k->status = STORM_HIP_OK, k->stepping = stepping;
  k->gram_schmidt = p->gram_schmidt;
  k->lag = p->check_lag > 0 ? p->check_lag : 4;
  if (k->lag > kStateRing - 1) k->lag = kStateRing - 1;
  switch (k->method) {
    case STORM_HIP_GMRES:
    case STORM_HIP_FGMRES: k->inner = (int)(p->num_inner_iterations > 0 ? p->num_inner_iterations : 50); break;
    case STORM_HIP_BICGSTAB_L: k->inner = (int)(p->num_inner_iterations > 0 ? p->num_inner_iterations : 2); break;
    case STORM_HIP_IDRS: k->inner = (int)(p->num_inner_iterations > 0 ? p->num_inner_iterations : 4); break;
    default: k->inner = 0;
  }
  if (k->method == STORM_HIP_BICGSTAB_L || k->method == STORM_HIP_IDRS)
    STORM_REQUIRE(k->inner <= 48, "krylov: num_inner_iterations = %d too large for this method (<= 48)", k->inner);
  if (k->method == STORM_HIP_IDRS)
    STORM_REQUIRE(c->comm == nullptr, "krylov: IDR(s) draws its shadow space with fill_randomly, single rank only");
  k->reset_prog(), k->red_pending = false;
  k->dp = nullptr;
  k->applies = k->pre_applies = 0;
  k->it_enqueued = 0;
  k->active = true;
  k->setup();
  if (!k->ok()) return k->status;
  // register file
  if (k->S_top > k->S_cap) {
    if (k->S) (void)hipFree(k->S);
    k->S = nullptr, k->S_cap = 0;
    HIP_TRY(hipMalloc((void **)&k->S, sizeof(double) * (size_t)k->S_top));
    k->S_cap = k->S_top;
  }
  HIP_TRY(hipMemsetAsync(k->S, 0, sizeof(double) * (size_t)k->S_top, c->stream));
  {
    static const double one = 1.0;
    HIP_TRY(hipMemcpyAsync(k->S + R_ONE, &one, sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  // solver state
  for (int i = 0; i < kStateRing; ++i) k->h_ring[i] = 0;
  if (history && !stepping) {
    HIP_TRY(hipMalloc((void **)&k->d_history, sizeof(double) * (size_t)(p->num_iterations + 1)));
    HIP_TRY(hipMemsetAsync(k->d_history, 0, sizeof(double) * (size_t)(p->num_iterations + 1), c->stream));
  }
  // (stepping: the caller owns the convergence decision)
  STORM_TRY(state_init(c, k->d_st, stepping ? 0.0 : p->absolute_error_tolerance, stepping ? 0.0 : p->relative_error_tolerance,
                       stepping ? (1LL << 62) : p->num_iterations, (history && !stepping) ? k->d_history : nullptr, k->d_ring));
  k->init();
  k->applies_after.assign(1, k->applies);
  k->pre_after.assign(1, k->pre_applies);
  k->dp = &k->d_st->done;
  return k->status;
}

int read_state(K *k) {
  return state_read(k->c, k->d_st, k->h_st);
}

typedef int (*fused_entry)(const storm_hip_op *, double, double, const storm_hip_vec *, storm_hip_vec *,
                           const storm_hip_solver_params *, storm_hip_solver_result *, double *);

}  // namespace

extern "C" {

int storm_hip_krylov_create(storm_hip_ctx *ctx, int method, storm_hip_krylov **out) {
  STORM_REQUIRE(ctx && out, "krylov_create: null argument");
  STORM_REQUIRE(method >= STORM_HIP_CG && method <= STORM_HIP_RICHARDSON, "krylov_create: unknown method %d", method);
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  auto *k = new storm_hip_krylov();
  k->c = ctx, k->method = method;
  if (!ctx->krylov_free.empty()) {  // what a destroyed engine of this context left (begin_solve writes all of it anew)
    const KrylovRes r = ctx->krylov_free.back();
    ctx->krylov_free.pop_back();
    k->d_st = r.d_st, k->h_st = r.h_st, k->h_ring = r.h_ring, k->d_ring = r.d_ring, k->S = r.S, k->S_cap = r.S_cap;
  } else {
    hipError_t e = hipMalloc((void **)&k->d_st, sizeof(SolverState));
    if (e == hipSuccess) e = hipMemset(k->d_st, 0, sizeof(SolverState));
    if (e == hipSuccess) e = hipHostMalloc((void **)&k->h_st, sizeof(SolverState), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&k->h_ring, sizeof(unsigned long long) * kStateRing, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&k->d_ring, k->h_ring, 0);
    if (e != hipSuccess) {
      if (k->d_st) (void)hipFree(k->d_st);
      if (k->h_st) (void)hipHostFree(k->h_st);
      if (k->h_ring) (void)hipHostFree(k->h_ring);
      delete k;
      HIP_TRY(e);
    }
  }
  *out = k;
  return STORM_HIP_OK;
}

int storm_hip_krylov_destroy(storm_hip_krylov *k) {
  if (!k) return STORM_HIP_OK;
  (void)hipSetDevice(k->c->device);
  (void)hipStreamSynchronize(k->c->stream);
  release_work(k);
  for (auto &e : k->ev) (void)hipEventDestroy(e);
  if (k->c->krylov_free.size() < 8) {  // (the stream is idle: nothing reads these any more)
    k->c->krylov_free.push_back(KrylovRes{k->d_st, k->h_st, k->h_ring, k->d_ring, k->S, k->S_cap});
  } else {
    if (k->S) (void)hipFree(k->S);
    (void)hipFree(k->d_st);
    (void)hipHostFree(k->h_st);
    (void)hipHostFree(k->h_ring);
  }
  delete k;
  return STORM_HIP_OK;
}

int storm_hip_krylov_set_operator(storm_hip_krylov *k, const storm_hip_op *op, double alpha, double beta) {
  STORM_REQUIRE(k && op, "krylov_set_operator: null argument");
  k->op = op, k->op_alpha = alpha, k->op_beta = beta, k->op_fn = nullptr, k->op_user = nullptr;
  return STORM_HIP_OK;
}

int storm_hip_krylov_set_operator_fn(storm_hip_krylov *k, storm_hip_apply_fn apply, void *user) {
  STORM_REQUIRE(k && apply, "krylov_set_operator_fn: null argument");
  k->op = nullptr, k->op_fn = apply, k->op_user = user;
  return STORM_HIP_OK;
}

int storm_hip_krylov_set_preconditioner_fn(storm_hip_krylov *k, storm_hip_apply_fn apply, void *user, int side) {
  STORM_REQUIRE(k, "krylov_set_preconditioner_fn: null solver");
  STORM_REQUIRE(side >= STORM_HIP_LEFT && side <= STORM_HIP_SYMMETRIC, "krylov: unknown preconditioner side %d", side);
  k->pre_fn = apply, k->pre_user = user, k->pre_diag = nullptr, k->side = side;
  return STORM_HIP_OK;
}

int storm_hip_krylov_set_preconditioner_diag(storm_hip_krylov *k, const storm_hip_vec *d, int side) {
  STORM_REQUIRE(k, "krylov_set_preconditioner_diag: null solver");
  STORM_REQUIRE(side >= STORM_HIP_LEFT && side <= STORM_HIP_SYMMETRIC, "krylov: unknown preconditioner side %d", side);
  STORM_REQUIRE(d == nullptr || d->ctx == k->c, "krylov: preconditioner diagonal belongs to another context");
  k->pre_fn = nullptr, k->pre_user = nullptr, k->pre_diag = d, k->side = side;
  return STORM_HIP_OK;
}

int storm_hip_krylov_set_real(storm_hip_krylov *k, const char *key, double value) {
  STORM_REQUIRE(k && key, "krylov_set_real: null argument");
  if (!strcmp(key, "relaxation_factor")) k->relaxation = value;
  else STORM_FAIL(STORM_HIP_E_INVALID, "krylov_set_real: unknown key '%s'", key);
  return STORM_HIP_OK;
}

static int krylov_solve_engine(storm_hip_krylov *k, const storm_hip_vec *b, storm_hip_vec *x,
                               const storm_hip_solver_params *params, storm_hip_solver_result *result, double *history,
                               int64_t *pre_applies);
namespace {
struct EngineSolveArgs {
  storm_hip_krylov *k;
  const storm_hip_vec *b;
  storm_hip_vec *x;
  const storm_hip_solver_params *params;
  storm_hip_solver_result *result;
  double *history;
  int64_t *pre_applies;
};
int run_engine_body(void *p) {
  const EngineSolveArgs &a = *static_cast<const EngineSolveArgs *>(p);
  return krylov_solve_engine(a.k, a.b, a.x, a.params, a.result, a.history, a.pre_applies);
}
}  // namespace

int storm_hip_krylov_solve(storm_hip_krylov *k, const storm_hip_vec *b, storm_hip_vec *x,
                           const storm_hip_solver_params *params, storm_hip_solver_result *result, double *history,
                           int64_t *pre_applies) {
  STORM_REQUIRE(k && result, "krylov_solve: null argument");
  STORM_TRY(lazy_sync(k->c));
  STORM_TRY(check_ready(k, b, x, params));
  storm_hip_ctx *c = k->c;
  HIP_TRY(hipSetDevice(c->device));
  // (the engine's GMRES runs its Gram-Schmidt chains as cooperative kernels where they fit: should one give up, x is
  //  restored and the solve re-run without them -- latency.hip, coop_solve_with_fallback)
  EngineSolveArgs a{k, b, x, params, result, history, pre_applies};
  int fb = 0;
  const bool fused_path = k->op != nullptr && !k->has_pre() && c->opt_generic_solvers == 0 &&
                          (k->method == STORM_HIP_CG || k->method == STORM_HIP_BICGSTAB ||
                           (k->method == STORM_HIP_GMRES && params->num_inner_iterations < kMaxMulti));
  if (fused_path) return krylov_solve_engine(k, b, x, params, result, history, pre_applies);  // (has its own fallback)
  const int st = coop_solve_with_fallback(c, x, run_engine_body, &a, &fb);
  if (st == STORM_HIP_OK) result->path_fallback = fb;
  return st;
}

static int krylov_solve_engine(storm_hip_krylov *k, const storm_hip_vec *b, storm_hip_vec *x,
                               const storm_hip_solver_params *params, storm_hip_solver_result *result, double *history,
                               int64_t *pre_applies) {
  storm_hip_ctx *c = k->c;
  // A stencil operator without preconditioner: CG / BiCGStab / GMRES have fused kernels (solvers.hip).
  if (k->op != nullptr && !k->has_pre() && c->opt_generic_solvers == 0) {
    fused_entry fused = k->method == STORM_HIP_CG         ? &storm_hip_solve_cg
                        : k->method == STORM_HIP_BICGSTAB ? &storm_hip_solve_bicgstab
                        : k->method == STORM_HIP_GMRES && params->num_inner_iterations < kMaxMulti
                            ? &storm_hip_solve_gmres  // (its state slab holds restarts below kMaxMulti)
                            : nullptr;
    if (fused != nullptr) {
      if (pre_applies) *pre_applies = 0;
      return fused(k->op, k->op_alpha, k->op_beta, b, x, params, result, history);
    }
  }
  int st = begin_solve(k, b, x, params, false, history);
  for (int64_t it = 0; st == STORM_HIP_OK && it < params->num_iterations; ++it) {
    k->iterate(it);
    st = k->status;
    if (st != STORM_HIP_OK) break;
    k->it_enqueued = it + 1;
    k->applies_after.push_back(k->applies);
    k->pre_after.push_back(k->pre_applies);
    // post a marker behind this iteration; look at the verdict of iteration it - lag
    st = ring_post(c, k->ev, it);
    if (st == STORM_HIP_OK && it >= k->lag) {
      bool stop = false;
      st = ring_wait(c, k->ev, k->h_ring, it - k->lag, &stop);
      if (stop) break;
    }
  }
  if (st == STORM_HIP_OK) st = read_state(k);
  if (st == STORM_HIP_OK) st = lat_check_gave_up(c);
  if (st == STORM_HIP_OK) {
    const int64_t iters = k->h_st->iteration;
    const int64_t a0 = k->applies, p0 = k->pre_applies;
    k->finalize(iters, true);
    st = k->status;
    const size_t at = (size_t)std::min<int64_t>(iters, (int64_t)k->applies_after.size() - 1);
    result->iterations = iters;
    result->absolute_error = k->h_st->absolute_error;
    result->relative_error = k->h_st->relative_error;
    result->initial_error = k->h_st->initial_error;
    result->converged = k->h_st->converged;
    result->num_applies = k->applies_after[at] + (k->applies - a0);
    if (pre_applies) *pre_applies = k->pre_after[at] + (k->pre_applies - p0);
    if (st == STORM_HIP_OK && history && k->d_history)
      HIP_TRY(hipMemcpy(history, k->d_history, sizeof(double) * (size_t)(iters + 1), hipMemcpyDeviceToHost));
    if (st == STORM_HIP_OK) HIP_TRY(hipStreamSynchronize(c->stream));
  }
  (void)hipStreamSynchronize(c->stream);
  release_work(k);
  if (st == STORM_HIP_OK) st = comm_check_error(c);  // (a transport's bounded wait gave up during this solve)
  return st;
}

int storm_hip_krylov_init(storm_hip_krylov *k, const storm_hip_vec *b, storm_hip_vec *x,
                          const storm_hip_solver_params *params, double *initial_error) {
  STORM_REQUIRE(k && initial_error, "krylov_init: null argument");
  STORM_TRY(lazy_sync(k->c));
  int st = begin_solve(k, b, x, params, true, nullptr);
  if (st == STORM_HIP_OK) st = read_state(k);
  if (st != STORM_HIP_OK) {
    release_work(k);
    return st;
  }
  *initial_error = k->h_st->initial_error;
  return STORM_HIP_OK;
}

int storm_hip_krylov_iterate(storm_hip_krylov *k, double *error) {
  STORM_REQUIRE(k && error, "krylov_iterate: null argument");
  STORM_REQUIRE(k->active && k->stepping, "krylov_iterate: no storm_hip_krylov_init before");
  HIP_TRY(hipSetDevice(k->c->device));
  STORM_TRY(lazy_sync(k->c));
  k->iterate(k->it_enqueued);
  if (!k->ok()) return k->status;
  k->it_enqueued += 1;
  STORM_TRY(read_state(k));
  *error = k->h_st->absolute_error;
  return STORM_HIP_OK;
}

int storm_hip_krylov_finalize(storm_hip_krylov *k) {
  STORM_REQUIRE(k, "krylov_finalize: null solver");
  STORM_REQUIRE(k->active && k->stepping, "krylov_finalize: no storm_hip_krylov_init before");
  HIP_TRY(hipSetDevice(k->c->device));
  STORM_TRY(lazy_sync(k->c));
  k->finalize(k->it_enqueued, false);
  const int st = k->status;
  (void)hipStreamSynchronize(k->c->stream);
  release_work(k);
  return st;
}

}  // extern "C"
