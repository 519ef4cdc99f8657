/*
 * The general Krylov engine driven from plain C (C99) with a CALLBACK operator -- the shape of the reference's only call
 * site, a lambda handed to a solver (Playground.cpp:151-167): the callback enqueues y = A x with library calls, the
 * solver loop around it stays on the device.  Every method of the engine on the 1-D Poisson known answer of SURVEY.md 8c
 * (64 unknowns, rows (-1, 2, -1), b = 1  =>  x[31] = 528; CG, CGS, TFQMR1 and GMRES reach it in 32 iterations, the counts
 * the reference's templates produce), with a diagonal preconditioner on either side, through the stepping interface
 * (init / iterate / finalize = the reference's protected hooks), and a callback that fails.
 *
 *   gcc -std=c99 -Iinclude tests/c/abi_krylov_callback.c -Lstormruler_amd -lstorm_hip -lm -o abi_krylov_callback
 * prints one JSON line per case; exit status 0 iff every known answer is met.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <storm_hip.h>

#define CHECK(call)                                                            \
  do {                                                                         \
    int st_ = (call);                                                          \
    if (st_ != STORM_HIP_OK) {                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, st_, storm_hip_last_error());   \
      return 2;                                                                \
    }                                                                          \
  } while (0)

typedef struct {
  const storm_hip_op *op;
  long calls;
  int fail_at; /* > 0: return an error from that call on */
} lambda_state;

/* y = A x, enqueue only.  (A = 1 * M + 0 * I of the CSR operator.) */
static int apply_lambda(void *user, storm_hip_vec *y, const storm_hip_vec *x) {
  lambda_state *s = (lambda_state *)user;
  s->calls++;
  if (s->fail_at > 0 && s->calls >= s->fail_at) return 7;
  return storm_hip_op_apply(s->op, 1.0, 0.0, x, y);
}

int main(void) {
  enum { N = 64 };
  int64_t row_ptr[N + 1], col[3 * N];
  double val[3 * N], ones[N], zeros[N], x_host[N], half[N];
  int64_t nnz = 0;
  for (int i = 0; i < N; ++i) {
    row_ptr[i] = nnz;
    if (i > 0) col[nnz] = i - 1, val[nnz++] = -1.0;
    col[nnz] = i, val[nnz++] = 2.0;
    if (i < N - 1) col[nnz] = i + 1, val[nnz++] = -1.0;
    ones[i] = 1.0, zeros[i] = 0.0, half[i] = 0.5; /* diag(A)^-1 */
  }
  row_ptr[N] = nnz;

  storm_hip_ctx *ctx = NULL;
  storm_hip_op *op = NULL;
  storm_hip_vec *b = NULL, *x = NULL, *dinv = NULL;
  CHECK(storm_hip_ctx_create(0, &ctx));
  CHECK(storm_hip_op_create_csr(ctx, N, 0, row_ptr, col, val, &op));
  CHECK(storm_hip_vec_create(ctx, N, 0, &b));
  CHECK(storm_hip_vec_create(ctx, N, 0, &x));
  CHECK(storm_hip_vec_create(ctx, N, 0, &dinv));
  CHECK(storm_hip_vec_upload(b, ones, N));
  CHECK(storm_hip_vec_upload(dinv, half, N));

  static const struct { int method; const char *name; int exact_its; int inner; } cases[] = {
      {STORM_HIP_CG, "cg", 32, 0},          {STORM_HIP_BICGSTAB, "bicgstab", 0, 0}, {STORM_HIP_GMRES, "gmres", 32, 50},
      {STORM_HIP_FGMRES, "fgmres", 32, 50}, {STORM_HIP_CGS, "cgs", 32, 0},          {STORM_HIP_TFQMR, "tfqmr", 0, 0},
      {STORM_HIP_TFQMR1, "tfqmr1", 32, 0},  {STORM_HIP_BICGSTAB_L, "bicgstabl", 0, 2}, {STORM_HIP_IDRS, "idrs", 0, 4}};
  int bad = 0;
  storm_hip_solver_params p;
  storm_hip_solver_result r;
  for (size_t c = 0; c < sizeof cases / sizeof cases[0]; ++c) {
    for (int side = -1; side <= STORM_HIP_RIGHT; ++side) { /* -1: no preconditioner */
      storm_hip_krylov *k = NULL;
      lambda_state lam = {op, 0, 0};
      int64_t n_pre = 0;
      CHECK(storm_hip_krylov_create(ctx, cases[c].method, &k));
      CHECK(storm_hip_krylov_set_operator_fn(k, apply_lambda, &lam));
      if (side >= 0) CHECK(storm_hip_krylov_set_preconditioner_diag(k, dinv, side));
      storm_hip_solver_params_default(&p);
      p.absolute_error_tolerance = 1e-10, p.relative_error_tolerance = 1e-12;
      p.num_inner_iterations = cases[c].inner;
      CHECK(storm_hip_vec_upload(x, zeros, N));
      storm_hip_rng_reset();
      CHECK(storm_hip_krylov_solve(k, b, x, &p, &r, NULL, &n_pre));
      CHECK(storm_hip_vec_download(x, x_host, N));
      const int ok = r.converged && fabs(x_host[31] - 528.0) < 1e-6 && (side >= 0 || cases[c].exact_its == 0 ||
                                                                          r.iterations == cases[c].exact_its) &&
                     r.num_applies <= lam.calls && (side < 0 ? n_pre == 0 : n_pre > 0);
      printf("{\"method\": \"%s\", \"side\": %d, \"iterations\": %lld, \"applies\": %lld, \"callback_entries\": %ld, "
             "\"pre_applies\": %lld, \"x31\": %.17g, \"ok\": %d}\n",
             cases[c].name, side, (long long)r.iterations, (long long)r.num_applies, lam.calls, (long long)n_pre,
             x_host[31], ok);
      bad += !ok;
      CHECK(storm_hip_krylov_destroy(k));
    }
  }
  { /* Richardson with the Jacobi diagonal = damped Jacobi: converges (slowly); only the residual is checked */
    storm_hip_krylov *k = NULL;
    lambda_state lam = {op, 0, 0};
    CHECK(storm_hip_krylov_create(ctx, STORM_HIP_RICHARDSON, &k));
    CHECK(storm_hip_krylov_set_operator_fn(k, apply_lambda, &lam));
    CHECK(storm_hip_krylov_set_preconditioner_diag(k, dinv, STORM_HIP_RIGHT));
    CHECK(storm_hip_krylov_set_real(k, "relaxation_factor", 1.0));
    storm_hip_solver_params_default(&p);
    p.num_iterations = 200, p.absolute_error_tolerance = 0.0, p.relative_error_tolerance = 0.0;
    CHECK(storm_hip_vec_upload(x, zeros, N));
    CHECK(storm_hip_krylov_solve(k, b, x, &p, &r, NULL, NULL));
    const int ok = r.iterations == 200 && !r.converged && r.absolute_error < r.initial_error && r.num_applies == 201;
    printf("{\"method\": \"richardson\", \"iterations\": %lld, \"abs\": %.6e, \"initial\": %.6e, \"ok\": %d}\n",
           (long long)r.iterations, r.absolute_error, r.initial_error, ok);
    bad += !ok;
    CHECK(storm_hip_krylov_destroy(k));
  }
  { /* the stepping interface: the caller owns the loop and the convergence decision (Solver.hpp:116-147) */
    storm_hip_krylov *k = NULL;
    lambda_state lam = {op, 0, 0};
    double err0 = 0.0, err = 0.0;
    int its = 0;
    CHECK(storm_hip_krylov_create(ctx, STORM_HIP_CG, &k));
    CHECK(storm_hip_krylov_set_operator_fn(k, apply_lambda, &lam));
    storm_hip_solver_params_default(&p);
    CHECK(storm_hip_vec_upload(x, zeros, N));
    CHECK(storm_hip_krylov_init(k, b, x, &p, &err0));
    for (err = err0; its < 2000 && !(err < 1e-10 || err / err0 < 1e-12); ++its) CHECK(storm_hip_krylov_iterate(k, &err));
    CHECK(storm_hip_krylov_finalize(k));
    CHECK(storm_hip_vec_download(x, x_host, N));
    const int ok = its == 32 && fabs(x_host[31] - 528.0) < 1e-6 && fabs(err0 - 8.0) < 1e-12 && lam.calls == 33;
    printf("{\"method\": \"cg/stepping\", \"iterations\": %d, \"initial_error\": %.17g, \"callback_entries\": %ld, \"ok\": %d}\n",
           its, err0, lam.calls, ok);
    bad += !ok;
    CHECK(storm_hip_krylov_destroy(k));
  }
  { /* a callback that fails aborts the solve with an error status; the library stays usable */
    storm_hip_krylov *k = NULL;
    lambda_state lam = {op, 0, 5};
    CHECK(storm_hip_krylov_create(ctx, STORM_HIP_CG, &k));
    CHECK(storm_hip_krylov_set_operator_fn(k, apply_lambda, &lam));
    storm_hip_solver_params_default(&p);
    CHECK(storm_hip_vec_upload(x, zeros, N));
    const int st = storm_hip_krylov_solve(k, b, x, &p, &r, NULL, NULL);
    const int ok = st == STORM_HIP_E_INVALID && strstr(storm_hip_last_error(), "callback") != NULL;
    printf("{\"method\": \"cg/failing-callback\", \"status\": %d, \"ok\": %d}\n", st, ok);
    bad += !ok;
    CHECK(storm_hip_krylov_destroy(k));
    CHECK(storm_hip_ctx_sync(ctx));
  }
  CHECK(storm_hip_vec_destroy(dinv));
  CHECK(storm_hip_vec_destroy(x));
  CHECK(storm_hip_vec_destroy(b));
  CHECK(storm_hip_op_destroy(op));
  CHECK(storm_hip_ctx_destroy(ctx));
  return bad ? 1 : 0;
}
