/*
 * The drop-in boundary used from plain C (C99, no C++ runtime on the caller's side): the 1-D Poisson known
 * answer of SURVEY.md 8c -- 64 unknowns, rows (-1, 2, -1), b = 1  =>  x[31] = 528 exactly, reached by CG in 32
 * iterations (the count the reference's own CgSolver template produces on this system).
 *
 *   gcc -std=c99 -Iinclude tests/c/abi_poisson1d.c -Lstormruler_amd -lstorm_hip -o abi_poisson1d
 * prints one JSON line; exit status 0 iff the known answer is met.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <storm_hip.h>

#define CHECK(call)                                                            \
  do {                                                                         \
    int st_ = (call);                                                          \
    if (st_ != STORM_HIP_OK) {                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, st_, storm_hip_last_error());   \
      return 2;                                                                \
    }                                                                          \
  } while (0)

int main(void) {
  enum { N = 64 };
  int64_t row_ptr[N + 1], col[3 * N];
  double val[3 * N], ones[N], x_host[N];
  int64_t nnz = 0;
  for (int i = 0; i < N; ++i) {
    row_ptr[i] = nnz;
    if (i > 0) col[nnz] = i - 1, val[nnz++] = -1.0;
    col[nnz] = i, val[nnz++] = 2.0;
    if (i < N - 1) col[nnz] = i + 1, val[nnz++] = -1.0;
    ones[i] = 1.0;
  }
  row_ptr[N] = nnz;

  storm_hip_ctx *ctx = NULL;
  storm_hip_op *op = NULL;
  storm_hip_vec *b = NULL, *x = NULL;
  CHECK(storm_hip_ctx_create(0, &ctx));
  CHECK(storm_hip_op_create_csr(ctx, N, 0, row_ptr, col, val, &op));
  CHECK(storm_hip_vec_create(ctx, N, 0, &b));
  CHECK(storm_hip_vec_create(ctx, N, 0, &x)); /* zero-initialised: x0 = 0 */
  CHECK(storm_hip_vec_upload(b, ones, N));

  storm_hip_solver_params p;
  storm_hip_solver_result r;
  storm_hip_solver_params_default(&p); /* the reference's defaults, Solver.hpp:66-72 */
  p.absolute_error_tolerance = 1e-10;
  p.relative_error_tolerance = 1e-12;
  CHECK(storm_hip_solve_cg(op, 1.0, 0.0, b, x, &p, &r, NULL)); /* A = 1 * M + 0 * I */
  CHECK(storm_hip_vec_download(x, x_host, N));

  printf("{\"iterations\": %lld, \"converged\": %d, \"x31\": %.17g, \"relative_error\": %.3e}\n",
         (long long)r.iterations, (int)r.converged, x_host[31], r.relative_error);
  const int ok = r.converged && r.iterations == 32 && fabs(x_host[31] - 528.0) < 1e-6;

  CHECK(storm_hip_vec_destroy(x));
  CHECK(storm_hip_vec_destroy(b));
  CHECK(storm_hip_op_destroy(op));
  CHECK(storm_hip_ctx_destroy(ctx));
  return ok ? 0 : 1;
}
