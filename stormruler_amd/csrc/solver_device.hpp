// Device-side pieces shared by the solver translation units (solvers.hip: the fused CG / BiCGStab / GMRES
// loops of a stencil operator; krylov.hip: the general Krylov engine): the reference's scalar helpers and the
// body of IterativeSolver::solve's loop, evaluated on the device against a SolverState.
#pragma once

#include "common.hpp"
#include "wave_device.hpp"

namespace storm {

// Crow/MathUtils.hpp:49-52
__device__ __forceinline__ double safe_divide(double x, double y) { return (y == 0.0) ? 0.0 : (x / y); }

// The body of the for loop in IterativeSolver::solve, Solver.hpp:132-140.
__device__ inline void advance(SolverState *st, double abs_err) {
  st->absolute_error = abs_err;
  st->relative_error = abs_err / st->initial_error;
  bool conv = false;
  conv |= (st->abs_tol > 0.0) && (st->absolute_error < st->abs_tol);
  conv |= (st->rel_tol > 0.0) && (st->relative_error < st->rel_tol);
  st->iteration += 1;
  if (st->history) st->history[st->iteration] = abs_err;
  if (conv) st->converged = 1;
  if (conv || st->iteration >= st->num_iterations) st->done = 1;
  // Tell the host (it polls this pinned ring `check_lag` iterations behind): one system-scope store of a word that
  // names its iteration -- the host needs no event behind the kernel to trust it (common.hpp ring_wait).
  if (st->done_ring)
    __hip_atomic_store(st->done_ring + (st->iteration - 1) % kStateRing,
                       ring_word(st->ring_gen, (unsigned long long)st->iteration, st->done != 0), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// After init(): Solver.hpp:122-128.
__device__ inline void begin(SolverState *st, double initial_error) {
  st->initial_error = initial_error;
  st->absolute_error = initial_error;
  st->relative_error = 0.0;
  st->iteration = 0;
  st->converged = 0;
  st->done = 0;
  if (st->history) st->history[0] = initial_error;
  if (st->abs_tol > 0.0 && initial_error < st->abs_tol) st->converged = 1, st->done = 1;
  if (st->num_iterations <= 0) st->done = 1;
  if (st->done && st->done_ring)  // no iterate() will run: every poll must see it
    for (int i = 0; i < kStateRing; ++i)
      __hip_atomic_store(st->done_ring + i, ring_word(st->ring_gen, kRingIterMask, true), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The device-side state of a GMRES cycle and the Givens update of Hessenberg column k with the beta recurrence
// (SolverGmres.hpp:176-191; sym_ortho: Crow/MathUtils.hpp:164-179), followed by the convergence rule.  One thread.
struct GmresDev {
  double *H, *beta, *cs, *sn;  // device arrays: (m+1) x m row-major, m+1, m, m
  int m;
};
__device__ inline void gmres_givens_update(SolverState *st, GmresDev g, int k, double hn) {
  const int m = g.m;
#define H_(i, j) g.H[(i) * m + (j)]
  H_(k + 1, k) = hn;
  for (int i = 0; i < k; ++i) {
    const double chi = g.cs[i] * H_(i, k) + g.sn[i] * H_(i + 1, k);
    H_(i + 1, k) = -g.sn[i] * H_(i, k) + g.cs[i] * H_(i + 1, k);
    H_(i, k) = chi;
  }
  const double a = H_(k, k), b = H_(k + 1, k);
  const double rr = hypot(a, b);
  double cs, sn;
  if (rr > 0.0) cs = a / rr, sn = b / rr;
  else cs = 1.0, sn = 0.0;
  g.cs[k] = cs, g.sn[k] = sn;
  H_(k, k) = cs * H_(k, k) + sn * H_(k + 1, k);
  H_(k + 1, k) = 0.0;
  g.beta[k + 1] = -sn * g.beta[k];
  g.beta[k] *= cs;
  advance(st, fabs(g.beta[k + 1]));
#undef H_
}

__device__ __forceinline__ double block_sum256(double v, double *lds4) {
  v = wave_sum_down(v);  // (the __shfl_down tree's order and bits, without the LDS crossbar: wave_device.hpp)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds4[wave] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

}  // namespace storm
