/*
 * oracle/storm_oracle_omp.c -- TEST / BENCH INFRASTRUCTURE ONLY (never imported by the product).
 *
 * "What the host CPU could do": an OpenMP-parallel CG on the same operator, reported by bench.py's
 * `cpu_baseline.parallel` beside the faithful single-threaded port (optional, clearly
 * labelled not-the-reference).  It is NOT the reference's algorithm order: the reference's face loop
 * (source_apps/playground/Playground.cpp:119-130) scatters into both cells of a face and cannot be split
 * across threads without races, and its reductions are sequential (Bittern/MatrixAlgorithms.hpp:191-205).
 * Here the stencil is applied in gather form over assembled CSR rows (one thread per row block) and the
 * dot products are OpenMP reductions; the CG recurrence itself is Solvers/SolverCg.hpp:86-126.  Results
 * agree with the sequential oracle to rounding (tests/test_oracle_kat.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <omp.h>

#define ORACLE_API __attribute__((visibility("default")))

static double pdot(int64_t n, const double *a, const double *b) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

static void pspmv(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val, double *y,
                  const double *x) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) s += val[k] * x[col[k]];
    y[i] = s;
  }
}

ORACLE_API int oracle_omp_max_threads(void) { return omp_get_max_threads(); }

/* `iterations` CG steps from x (tolerances off: a timing sample); returns the final residual norm. */
ORACLE_API double oracle_omp_cg(int64_t n, const int64_t *row_ptr, const int32_t *col, const double *val,
                                double *x, const double *b, int64_t iterations, int threads) {
  if (threads > 0) omp_set_num_threads(threads);
  double *p = (double *)malloc(sizeof(double) * (size_t)n), *r = (double *)malloc(sizeof(double) * (size_t)n),
         *z = (double *)malloc(sizeof(double) * (size_t)n);
  pspmv(n, row_ptr, col, val, r, x);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    r[i] = b[i] - r[i];
    p[i] = r[i];
  }
  double gamma = pdot(n, r, r);
  for (int64_t it = 0; it < iterations; ++it) {
    pspmv(n, row_ptr, col, val, z, p);
    const double pz = pdot(n, p, z);
    const double alpha = pz == 0.0 ? 0.0 : gamma / pz;
    double g2 = 0.0;
#pragma omp parallel for reduction(+ : g2) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      x[i] += alpha * p[i];
      r[i] -= alpha * z[i];
      g2 += r[i] * r[i];
    }
    const double beta = gamma == 0.0 ? 0.0 : g2 / gamma;
    gamma = g2;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];
  }
  free(p), free(r), free(z);
  return sqrt(gamma);
}

/*
 * The bench sample: the n^3 Dirichlet box of SURVEY.md 8d (A = -L: off-diagonals -1/h^2, diagonal
 * (#interior faces + 2 #wall faces)/h^2), b = 1, x0 = 0, built here with first-touch placement (every array is
 * initialised by the thread block that later streams it -- it matters on a many-socket host).  Runs
 * `iterations` CG steps, writes the elapsed seconds of the iteration loop (build excluded) and returns |r|.
 */
ORACLE_API double oracle_omp_cg_box(int n, int64_t iterations, int threads, double *seconds) {
  if (threads > 0) omp_set_num_threads(threads);
  const int64_t N = (int64_t)n * n * n, n2 = (int64_t)n * n;
  const double w = (double)n * (double)n;  /* 1 / h^2 */
  int64_t *row_ptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N + 1));
  int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(7 * N));
  double *val = (double *)malloc(sizeof(double) * (size_t)(7 * N));
  double *x = (double *)malloc(sizeof(double) * (size_t)N), *b = (double *)malloc(sizeof(double) * (size_t)N);
  double *p = (double *)malloc(sizeof(double) * (size_t)N), *r = (double *)malloc(sizeof(double) * (size_t)N);
  double *z = (double *)malloc(sizeof(double) * (size_t)N);
  /* fixed 7 slots per row (unused ones carry weight 0 and the row's own column): rows stay independent */
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < N; ++c) {
    const int i = (int)(c % n), j = (int)((c / n) % n), k = (int)(c / n2);
    int64_t at = 7 * c;
    row_ptr[c] = at;
    double diag = 0.0;
    const int64_t nb[6] = {c - n2, c - n, c - 1, c + 1, c + n, c + n2};
    const int ok[6] = {k > 0, j > 0, i > 0, i < n - 1, j < n - 1, k < n - 1};
    for (int q = 0; q < 6; ++q) {
      if (ok[q]) col[at] = (int32_t)nb[q], val[at] = -w, diag += w;
      else col[at] = (int32_t)c, val[at] = 0.0, diag += 2.0 * w;
      ++at;
    }
    col[at] = (int32_t)c, val[at] = diag;
    x[c] = 0.0, b[c] = 1.0, p[c] = 0.0, r[c] = 0.0, z[c] = 0.0;
  }
  row_ptr[N] = 7 * N;
  pspmv(N, row_ptr, col, val, r, x);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    r[i] = b[i] - r[i];
    p[i] = r[i];
  }
  double gamma = pdot(N, r, r);
  const double t0 = omp_get_wtime();
  for (int64_t it = 0; it < iterations; ++it) {
    pspmv(N, row_ptr, col, val, z, p);
    const double pz = pdot(N, p, z);
    const double alpha = pz == 0.0 ? 0.0 : gamma / pz;
    double g2 = 0.0;
#pragma omp parallel for reduction(+ : g2) schedule(static)
    for (int64_t i = 0; i < N; ++i) {
      x[i] += alpha * p[i];
      r[i] -= alpha * z[i];
      g2 += r[i] * r[i];
    }
    const double beta = gamma == 0.0 ? 0.0 : g2 / gamma;
    gamma = g2;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) p[i] = r[i] + beta * p[i];
  }
  if (seconds) *seconds = omp_get_wtime() - t0;
  free(row_ptr), free(col), free(val), free(x), free(b), free(p), free(r), free(z);
  return sqrt(gamma);
}
