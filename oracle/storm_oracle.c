/*
 * storm_oracle.c -- CPU restatement of the StormRuler Krylov hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it, and only as the
 * checker / the timed CPU baseline.  Nothing under stormruler_amd/ or include/
 * links, imports or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - BLAS-1 (dot / norm_2 / axpy-type expressions): PINNED by the reference's
 *     own unit tests tests/unit/BitternReductions.cpp:59-76,99-115 and
 *     tests/unit/BitternMath.cpp:136-189 (tests/test_oracle_kat.py).
 *   - stencil apply and the CG / BiCGStab / GMRES loops: PARITY UNPINNED by the
 *     reference's own tests (it has none for Storm::Solvers / stormDivGrad) and
 *     the reference cannot be built in this image without stand-ins for
 *     spdlog/fmt (Storm/Base.hpp -> Crow/Base/Log.hpp:23-24), which this repo
 *     does not write.  The restatement is cross-checked against the values the
 *     survey recorded from the reference in BASELINE.md section 2 (iteration
 *     counts, x[centre]) and against closed-form / scipy answers.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).  Arithmetic order is kept exactly as in the reference:
 * sequential left-to-right sums, one rounding per written operation; compile
 * with -ffp-contract=off so no FMA is formed that the source does not spell.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* Scalar helpers.                                                           */

/* source/Storm/Crow/MathUtils.hpp:49-52 -- y == 0 ? 0 : x / y. */
ORACLE_API double oracle_safe_divide(double x, double y) {
  return (y == 0.0) ? 0.0 : (x / y);
}

/* source/Storm/Crow/MathUtils.hpp:164-179 -- Givens rotation via hypot. */
ORACLE_API void oracle_sym_ortho(double a, double b, double *cs, double *sn,
                                 double *rr) {
  *rr = hypot(a, b);
  if (*rr > 0.0) {
    *cs = a / *rr, *sn = b / *rr;
  } else {
    *cs = 1.0, *sn = 0.0;
  }
}

/* ------------------------------------------------------------------------ */
/* BLAS-1: the Bittern element loops a solver statement lowers to.           */
/* source/Storm/Bittern/MatrixAlgorithms.hpp:58-81 (matrix_for_each, one     */
/* sequential loop over rows; NumVars == 1 so cols == 1, Field.hpp:77-79).   */

/* ORACLE_SUM_ORDER selects how the two reductions below add their n terms.   */
/*   0 (every build the tests and the bench load): the reference's order --    */
/*     init 0.0, strictly sequential, left to right;                           */
/*   1, 2: INVESTIGATION builds only (oracle/Makefile `variants`, loaded by    */
/*     tools/make_full_size_fixtures.py to measure how far a solver's          */
/*     iteration count moves with the summation order and nothing else):       */
/*     1 = pairwise tree over blocks of 256 (the shape of a GPU reduction),    */
/*     2 = one 80-bit long double accumulator (a correctly rounded sum for     */
/*         all practical purposes).                                            */
#ifndef ORACLE_SUM_ORDER
#define ORACLE_SUM_ORDER 0
#endif
#if ORACLE_SUM_ORDER == 1
static double pairwise_products(int64_t n, const double *a, const double *b) {
  if (n <= 256) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s = s + a[i] * b[i];
    return s;
  }
  const int64_t h = (n / 2 + 255) & ~(int64_t)255;
  return pairwise_products(h, a, b) + pairwise_products(n - h, a + h, b + h);
}
#endif
static double sum_of_products(int64_t n, const double *a, const double *b) {
#if ORACLE_SUM_ORDER == 0
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s = s + a[i] * b[i];
  return s;
#elif ORACLE_SUM_ORDER == 1
  return pairwise_products(n, a, b);
#else
  long double s = 0.0L;
  for (int64_t i = 0; i < n; ++i) s = s + (long double)(a[i] * b[i]);
  return (double)s;
#endif
}
ORACLE_API int oracle_sum_order(void) { return ORACLE_SUM_ORDER; }

/* dot_product: MatrixAlgorithms.hpp:310-317 -> reduce :191-205, init 0.0,   */
/* strictly sequential; real DotProduct = a * b (Crow/MathUtils.hpp:90-95).  */
ORACLE_API double oracle_dot(int64_t n, const double *a, const double *b) {
  return sum_of_products(n, a, b);
}

/* norm_2: MatrixAlgorithms.hpp:262-270: sqrt(sum |a_i|^2), AbsSquared =     */
/* real(a * conj(a)) (Crow/FunctionalUtils.hpp:488-496).                     */
ORACLE_API double oracle_norm2(int64_t n, const double *a) {
  return sqrt(sum_of_products(n, a, a));
}

/* sum / norm_1 / norm_inf: MatrixAlgorithms.hpp:214-300 (used by the KATs). */
ORACLE_API double oracle_sum(int64_t n, const double *a) {
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s = s + a[i];
  return s;
}
ORACLE_API double oracle_norm1(int64_t n, const double *a) {
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s = s + fabs(a[i]);
  return s;
}
ORACLE_API double oracle_norm_inf(int64_t n, const double *a) {
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s = fmax(s, fabs(a[i]));
  return s;
}

/* y <<= x            MatrixAlgorithms.hpp:120-124 */
ORACLE_API void oracle_copy(int64_t n, double *y, const double *x) {
  for (int64_t i = 0; i < n; ++i) y[i] = x[i];
}
/* fill_with(y, v)    (ADL hook, Solver.hpp:281) */
ORACLE_API void oracle_fill(int64_t n, double *y, double v) {
  for (int64_t i = 0; i < n; ++i) y[i] = v;
}
/* y += a * x         MatrixTarget.hpp:108-113 with MatrixMath.hpp:247-256 */
ORACLE_API void oracle_axpy(int64_t n, double *y, double a, const double *x) {
  for (int64_t i = 0; i < n; ++i) y[i] += a * x[i];
}
/* y -= a * x         MatrixTarget.hpp:114-119 */
ORACLE_API void oracle_axmy(int64_t n, double *y, double a, const double *x) {
  for (int64_t i = 0; i < n; ++i) y[i] -= a * x[i];
}
/* y <<= x + b * y    SolverCg.hpp:123 */
ORACLE_API void oracle_xpay(int64_t n, double *y, const double *x, double b) {
  for (int64_t i = 0; i < n; ++i) y[i] = x[i] + b * y[i];
}
/* p <<= r + b * (p - w * v)   SolverBiCgStab.hpp:119 */
ORACLE_API void oracle_bicg_p(int64_t n, double *p, const double *r, double b,
                              double w, const double *v) {
  for (int64_t i = 0; i < n; ++i) p[i] = r[i] + b * (p[i] - w * v[i]);
}
/* r <<= b - r        Operator.hpp:98 */
ORACLE_API void oracle_sub_from(int64_t n, double *r, const double *b) {
  for (int64_t i = 0; i < n; ++i) r[i] = b[i] - r[i];
}
/* y /= s ; y *= s    SolverGmres.hpp:88,162,242 (plain, not safe, divide) */
ORACLE_API void oracle_div_scalar(int64_t n, double *y, double s) {
  for (int64_t i = 0; i < n; ++i) y[i] /= s;
}
ORACLE_API void oracle_mul_scalar(int64_t n, double *y, double s) {
  for (int64_t i = 0; i < n; ++i) y[i] *= s;
}
/* out <<= a + s * (b - c)   tests/unit/BitternMath.cpp:146-151 (KAT expr-1) */
ORACLE_API void oracle_expr1(int64_t n, double *out, const double *a, double s,
                             const double *b, const double *c) {
  for (int64_t i = 0; i < n; ++i) out[i] = a[i] + s * (b[i] - c[i]);
}

/* out <<= (s * a) / b, or s / b when a == NULL: the quotient nodes of          */
/* Bittern/MatrixMath.hpp:261-265 (scalar / mat) and :298-302 (mat1 / mat2);   */
/* KATs expr-3 / expr-4, tests/unit/BitternMath.cpp:160-171.                   */
ORACLE_API void oracle_vdiv(int64_t n, double *out, double s, const double *a, const double *b) {
  for (int64_t i = 0; i < n; ++i) out[i] = a ? (s * a[i]) / b[i] : s / b[i];
}

/* ------------------------------------------------------------------------ */
/* The face graph: what stormDivGrad reads through the Mallard accessors     */
/* (Mesh.hpp:240-282 FaceView::inner_cell/outer_cell/area, :290-323          */
/* CellView::volume/center; interior_faces() = label-0 range, :453-455).     */

typedef struct oracle_mesh {
  int64_t n_cells, n_faces, n_bfaces;
  int32_t dim;
  const int64_t *inner;   /* [F] face -> inner cell (lower id side)          */
  const int64_t *outer;   /* [F] face -> outer cell                          */
  const double *area;     /* [F]                                             */
  const double *center;   /* [N * dim] cell centres                          */
  const double *volume;   /* [N]                                             */
  /* Dirichlet boundary faces (ghost-state pattern of             */
  /* Feathers/ConvectionScheme.hpp:95-106).  The ghost value sits at the     */
  /* face centre, i.e. at distance |x_f - x_c| from the cell centre.         */
  const int64_t *b_cell;  /* [B] boundary face -> its single (inner) cell    */
  const double *b_area;   /* [B]                                             */
  const double *b_center; /* [B * dim] face centres                          */
  const double *b_ghost;  /* [B] ghost values, or NULL for homogeneous 0     */
} oracle_mesh;

/* length(a - b): MatrixAlgorithms.hpp:303-305 -> norm_2 :262-270, i.e.      */
/* sqrt(0 + d0*d0 + d1*d1 (+ d2*d2)) summed left to right.                   */
static inline double center_dist(const double *a, const double *b, int dim) {
  double s = 0.0;
  for (int k = 0; k < dim; ++k) {
    const double d = a[k] - b[k];
    s = s + d * d;
  }
  return sqrt(s);
}

/*
 * u += dt * div(grad(c))      source_apps/playground/Playground.cpp:115-131
 *
 *   for each interior face (in face order):
 *     flux = dt * (c[out] - c[in]) / length(center(out) - center(in))   :126-127
 *     u[in]  += (area / volume(in))  * flux                             :128
 *     u[out] -= (area / volume(out)) * flux                             :129
 *
 * Geometry is recomputed on every apply, exactly as the reference does.
 * Boundary faces: the reference loop visits interior_faces() only (:119), a
 * pure-Neumann operator.  The Poisson configs need Dirichlet walls, so a
 * second loop adds the flux to a ghost state g_b held at the face centre
 * (loop shape of Feathers/ConvectionScheme.hpp:95-106); with n_bfaces == 0
 * this function is the reference stencil verbatim.
 */
ORACLE_API void oracle_divgrad(const oracle_mesh *m, double *u, double dt,
                               const double *c) {
  const int dim = m->dim;
  for (int64_t f = 0; f < m->n_faces; ++f) {
    const int64_t ci = m->inner[f], co = m->outer[f];
    const double flux =
        dt * (c[co] - c[ci]) /
        center_dist(m->center + co * dim, m->center + ci * dim, dim);
    u[ci] += (m->area[f] / m->volume[ci]) * flux;
    u[co] -= (m->area[f] / m->volume[co]) * flux;
  }
  for (int64_t b = 0; b < m->n_bfaces; ++b) {
    const int64_t ci = m->b_cell[b];
    const double g = m->b_ghost ? m->b_ghost[b] : 0.0;
    const double flux =
        dt * (g - c[ci]) /
        center_dist(m->b_center + b * dim, m->center + ci * dim, dim);
    u[ci] += (m->b_area[b] / m->volume[ci]) * flux;
  }
}

/*
 * First-order upwind convection, u += dt * div(v c), constant velocity v.
 * Loop shape of Feathers/ConvectionScheme.hpp:80-107: interior faces (:83-92)
 * take the upwind cell's state, flux added to inner / subtracted from outer
 * scaled by area/volume; boundary faces (:95-106) use the ghost state when the
 * flow enters.  `normal` is the unit normal pointing inner -> outer, taken
 * from the cell centres as in stormDivGrad.  (SURVEY 8f rank 1 operator; the
 * reference has no scalar convection definition -- Playground.cpp:161-165 is
 * a commented-out call site -- so this is the build's own definition.)
 */
ORACLE_API void oracle_convection(const oracle_mesh *m, double *u, double dt,
                                  const double *c, const double *vel) {
  const int dim = m->dim;
  for (int64_t f = 0; f < m->n_faces; ++f) {
    const int64_t ci = m->inner[f], co = m->outer[f];
    const double *xo = m->center + co * dim, *xi = m->center + ci * dim;
    const double d = center_dist(xo, xi, dim);
    double vn = 0.0;
    for (int k = 0; k < dim; ++k) vn = vn + vel[k] * ((xo[k] - xi[k]) / d);
    const double flux = dt * (vn > 0.0 ? vn * c[ci] : vn * c[co]);
    u[ci] += (m->area[f] / m->volume[ci]) * flux;   /* ConvectionScheme.hpp:90 */
    u[co] -= (m->area[f] / m->volume[co]) * flux;   /* ConvectionScheme.hpp:91 */
  }
  for (int64_t b = 0; b < m->n_bfaces; ++b) {
    const int64_t ci = m->b_cell[b];
    const double *xf = m->b_center + b * dim, *xi = m->center + ci * dim;
    const double d = center_dist(xf, xi, dim);
    double vn = 0.0;
    for (int k = 0; k < dim; ++k) vn = vn + vel[k] * ((xf[k] - xi[k]) / d);
    const double g = m->b_ghost ? m->b_ghost[b] : 0.0;
    const double flux = dt * (vn > 0.0 ? vn * c[ci] : vn * g);
    u[ci] += (m->b_area[b] / m->volume[ci]) * flux;  /* ConvectionScheme.hpp:104 */
  }
}

/* ------------------------------------------------------------------------ */
/* Operators.  `Operator::mul(y, x)`  Solvers/Operator.hpp:74.               */

typedef void (*oracle_apply_fn)(void *ctx, double *y, const double *x);

/*
 * The face-graph operator used by every config:
 *     y = beta * x + alpha * L(x)  [ + conv * C_v(x) ]
 * built the way the playground lambda builds its operator
 * (Playground.cpp:153-167): y <<= beta * x, then stormDivGrad(y, alpha, x).
 * Poisson:   alpha = -1, beta = 0.  Helmholtz: beta = 1, alpha = -kappa.
 * Conv-diff: alpha = -nu, beta = 0, conv = 1 (A = -nu L + C(v)).
 */
typedef struct oracle_stencil_op {
  const oracle_mesh *mesh;
  double alpha, beta;
  double conv;          /* 0 => no convection term */
  double vel[3];
} oracle_stencil_op;

ORACLE_API void oracle_stencil_apply(void *ctx, double *y, const double *x) {
  const oracle_stencil_op *op = (const oracle_stencil_op *)ctx;
  const int64_t n = op->mesh->n_cells;
  for (int64_t i = 0; i < n; ++i) y[i] = op->beta * x[i];
  oracle_divgrad(op->mesh, y, op->alpha, x);
  if (op->conv != 0.0) oracle_convection(op->mesh, y, op->conv, x, op->vel);
}

/* Plain CSR operator (1-D KATs, cross-checks).  y_i = sum_k val_k x[col_k]. */
typedef struct oracle_csr_op {
  int64_t n;
  const int64_t *row_ptr, *col;
  const double *val;
} oracle_csr_op;

ORACLE_API void oracle_csr_apply(void *ctx, double *y, const double *x) {
  const oracle_csr_op *op = (const oracle_csr_op *)ctx;
  for (int64_t i = 0; i < op->n; ++i) {
    double s = 0.0;
    for (int64_t k = op->row_ptr[i]; k < op->row_ptr[i + 1]; ++k)
      s = s + op->val[k] * x[op->col[k]];
    y[i] = s;
  }
}

/*
 * INVESTIGATION operator (tools/bicgstab_draw_study.py only; never loaded by a
 * parity check): the SAME operator in the arithmetic form the HIP kernels use
 * (stormruler_amd/csrc/spmv.hip) -- weights divided once at build time,
 * gather by rows in face order, difference form, the products contracted
 * into FMAs as hipcc does:
 *     acc = fma(w_ik, x[col_ik] - x_i, acc)           slot by slot
 *     y_i = fma(alpha, fma(ext_i, x_i, acc), beta * x_i)
 * plus an optional seeded perturbation of the result by at most one unit in the
 * last place (seed != 0: each y_i moves to its upper / lower neighbour with
 * probability 1/4 each), which is the size of the difference between any two
 * correctly rounded evaluation orders of the same row.
 * Used to measure how far BiCGStab's iteration count at 256^3 moves when
 * nothing but last-place roundings of the apply change.
 */
typedef struct oracle_gather_op {
  int64_t n;
  int32_t width;        /* slots per row; col < 0 marks an absent slot       */
  const int64_t *col;   /* [n * width] */
  const double *w;      /* [n * width] */
  const double *ext;    /* [n] */
  double alpha, beta;
  uint64_t seed;        /* 0: no perturbation */
  uint64_t applies;     /* counts applies (decorrelates the perturbations)   */
} oracle_gather_op;

static inline uint64_t gather_mix(uint64_t z) { /* splitmix64 finaliser */
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

ORACLE_API void oracle_gather_apply(void *ctx, double *y, const double *x) {
  oracle_gather_op *op = (oracle_gather_op *)ctx;
  const int W = op->width;
  const uint64_t salt = gather_mix(op->seed ^ (0x9e3779b97f4a7c15ull * (op->applies + 1)));
  op->applies++;
  for (int64_t i = 0; i < op->n; ++i) {
    const double xi = x[i];
    double acc = 0.0;
    for (int k = 0; k < W; ++k) {
      const int64_t c = op->col[i * W + k];
      if (c >= 0) acc = fma(op->w[i * W + k], x[c] - xi, acc);
    }
    double v = fma(op->alpha, fma(op->ext[i], xi, acc), op->beta * xi);
    if (op->seed != 0) {
      const uint64_t h = gather_mix(salt + (uint64_t)i) >> 62; /* 0..3 */
      if (h == 0) v = nextafter(v, INFINITY);
      else if (h == 1) v = nextafter(v, -INFINITY);
    }
    y[i] = v;
  }
}

/* Operator::Residual  Operator.hpp:95-99: mul(r, x); r <<= b - r. */
static void op_residual(oracle_apply_fn apply, void *ctx, int64_t n, double *r,
                        const double *b, const double *x) {
  apply(ctx, r, x);
  oracle_sub_from(n, r, b);
}

/* ------------------------------------------------------------------------ */
/* Solvers.                                                                  */

/* Public knobs of IterativeSolver, Solvers/Solver.hpp:66-72,158-159. */
typedef struct oracle_params {
  int64_t num_iterations;           /* default 2000 */
  double absolute_error_tolerance;  /* default 1e-6 */
  double relative_error_tolerance;  /* default 1e-6 */
  int64_t num_inner_iterations;     /* GMRES restart, default 50 */
  double relaxation_factor;         /* Richardson, default 1e-4 (SolverRichardson.hpp:45) */
} oracle_params;

typedef struct oracle_result {
  int64_t iterations;  /* IterativeSolver::iteration after solve()        */
  double absolute_error, relative_error, initial_error;
  int32_t converged;
  int64_t num_applies; /* operator applications (init + iterations)       */
} oracle_result;

typedef struct solver_vt {
  double (*init)(void *s, const double *x, const double *b);
  double (*iterate)(void *s, double *x, const double *b);
  void (*finalize)(void *s, double *x, const double *b);
} solver_vt;

typedef struct solver_base {
  oracle_apply_fn apply;
  void *op;
  int64_t n;
  int64_t iteration; /* IterativeSolver::iteration, Solver.hpp:66 */
  int64_t applies;
  double *history;   /* optional [num_iterations + 1] residual norms */
  oracle_apply_fn pre; /* IterativeSolver::pre_op (Solver.hpp:75), or NULL */
  void *pre_ctx;
  int side;          /* IterativeSolver::pre_side (Solver.hpp:74): 0 Left, 1 Right, 2 Symmetric */
  int64_t pre_applies;
} solver_base;

/* The preconditioner of the NEXT solve (pre_op / pre_side are members the caller sets before
 * solve(), Solver.hpp:74-75); every oracle_solve_* picks it up and clears it. */
static oracle_apply_fn g_pre = NULL;
static void *g_pre_ctx = NULL;
static int g_pre_side = 1;
static int64_t g_last_pre_applies = 0;
ORACLE_API void oracle_set_preconditioner(oracle_apply_fn pre, void *pre_ctx, int side) {
  g_pre = pre, g_pre_ctx = pre_ctx, g_pre_side = side;
}
ORACLE_API int64_t oracle_last_pre_applies(void) { return g_last_pre_applies; }
static void take_preconditioner(solver_base *sb) {
  sb->pre = g_pre, sb->pre_ctx = g_pre_ctx, sb->side = g_pre_side, sb->pre_applies = 0;
  g_pre = NULL, g_pre_ctx = NULL, g_pre_side = 1;
}
#define LEFT_PRE(sb) ((sb)->pre != NULL && (sb)->side == 0)
#define RIGHT_PRE(sb) ((sb)->pre != NULL && (sb)->side == 1)

static void op_mul(solver_base *s, double *y, const double *x) {
  s->apply(s->op, y, x);
  s->applies++;
}
static void pre_mul(solver_base *s, double *y, const double *x) {
  s->pre(s->pre_ctx, y, x);
  s->pre_applies++;
}
/* The three-way dispatch every preconditioned solver body repeats (e.g. SolverBiCgStab.hpp:134-137),
 * with the chained `mul(z, y, other, x)` of Operator.hpp:82-88 (other.mul(y, x); mul(z, y)):
 *   left : z = P(y = A x);   right: z = A(y = P x);   else: z = A x.   y may alias x only on the left. */
static void side_mul(solver_base *s, double *z, double *y, const double *x) {
  if (LEFT_PRE(s)) {
    op_mul(s, y, x);
    pre_mul(s, z, y);
  } else if (RIGHT_PRE(s)) {
    pre_mul(s, y, x);
    op_mul(s, z, y);
  } else {
    op_mul(s, z, x);
  }
}

/*
 * IterativeSolver::solve   Solvers/Solver.hpp:116-147.
 *  - initial_error = init(); absolute_error = initial_error;
 *  - early exit iff abs_tol > 0 && abs_err < abs_tol (finalize, return true)
 *  - for (iteration = 0; !converged && iteration < num_iterations; ++iteration)
 *      abs = iterate(); rel = abs / initial;
 *      converged |= abs_tol > 0 && abs < abs_tol;
 *      converged |= rel_tol > 0 && rel < rel_tol;
 *  - finalize; `iteration` ends as the number of iterate() calls.
 */
static void iterative_solve(solver_base *sb, const solver_vt *vt, void *s,
                            double *x, const double *b, const oracle_params *p,
                            oracle_result *res) {
  sb->applies = 0;
  const double initial_error = vt->init(s, x, b);
  res->initial_error = initial_error;
  res->absolute_error = initial_error;
  res->relative_error = 0.0;
  res->iterations = 0;
  if (sb->history) sb->history[0] = initial_error;
  if (p->absolute_error_tolerance > 0.0 &&
      res->absolute_error < p->absolute_error_tolerance) {
    if (vt->finalize) vt->finalize(s, x, b);
    res->converged = 1;
    res->num_applies = sb->applies;
    g_last_pre_applies = sb->pre_applies;
    return;
  }
  int converged = 0;
  for (sb->iteration = 0; !converged && (sb->iteration < p->num_iterations);
       ++sb->iteration) {
    res->absolute_error = vt->iterate(s, x, b);
    res->relative_error = res->absolute_error / initial_error;
    if (sb->history) sb->history[sb->iteration + 1] = res->absolute_error;
    converged |= (p->absolute_error_tolerance > 0.0) &&
                 (res->absolute_error < p->absolute_error_tolerance);
    converged |= (p->relative_error_tolerance > 0.0) &&
                 (res->relative_error < p->relative_error_tolerance);
  }
  if (vt->finalize) vt->finalize(s, x, b);
  res->iterations = sb->iteration;
  res->converged = converged;
  res->num_applies = sb->applies;
  g_last_pre_applies = sb->pre_applies;
}

/* Field::assign(other, copy) ignores `copy` and value-initialises a new     */
/* field: Feathers/Field.hpp:82-84.  calloc reproduces the zero fill.        */
static double *new_vec(int64_t n) {
  return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
}

/* ---- CG: Solvers/SolverCg.hpp:47-128 (both branches; CG ignores pre_side) - */
typedef struct cg_state {
  solver_base b;
  double gamma;
  double *p, *r, *z;
} cg_state;

/* SolverCg.hpp:54-84 */
static double cg_init(void *sv, const double *x, const double *b) {
  cg_state *s = (cg_state *)sv;
  const int64_t n = s->b.n;
  s->p = new_vec(n), s->r = new_vec(n), s->z = new_vec(n); /* :57-59 */
  s->b.applies++;
  op_residual(s->b.apply, s->b.op, n, s->r, b, x);          /* :75 */
  if (s->b.pre) {                                           /* :76-79 */
    pre_mul(&s->b, s->z, s->r);
    oracle_copy(n, s->p, s->z);
    s->gamma = oracle_dot(n, s->r, s->z);
    return oracle_norm2(n, s->r);                           /* :85 */
  }
  oracle_copy(n, s->p, s->r);                               /* :81 */
  s->gamma = oracle_dot(n, s->r, s->r);                     /* :82 */
  return sqrt(s->gamma);                                    /* :85 */
}
/* SolverCg.hpp:86-126 */
static double cg_iterate(void *sv, double *x, const double *b) {
  (void)b;
  cg_state *s = (cg_state *)sv;
  const int64_t n = s->b.n;
  op_mul(&s->b, s->z, s->p);                                              /* :96 */
  const double alpha = oracle_safe_divide(s->gamma, oracle_dot(n, s->p, s->z)); /* :97 */
  oracle_axpy(n, x, alpha, s->p);                                         /* :98 */
  oracle_axmy(n, s->r, alpha, s->z);                                      /* :99 */
  const double gamma_bar = s->gamma;                                      /* :110 */
  if (s->b.pre) {                                                         /* :111-113 */
    pre_mul(&s->b, s->z, s->r);
    s->gamma = oracle_dot(n, s->r, s->z);
    const double beta_p = oracle_safe_divide(s->gamma, gamma_bar);        /* :122 */
    oracle_xpay(n, s->p, s->z, beta_p);                                   /* :123 */
    return oracle_norm2(n, s->r);                                         /* :125 */
  }
  s->gamma = oracle_dot(n, s->r, s->r);                                   /* :115 */
  const double beta = oracle_safe_divide(s->gamma, gamma_bar);            /* :122 */
  oracle_xpay(n, s->p, s->r, beta);                                       /* :123 */
  return sqrt(s->gamma);                                                  /* :125 */
}

ORACLE_API void oracle_solve_cg(oracle_apply_fn apply, void *op, int64_t n,
                                double *x, const double *b,
                                const oracle_params *p, oracle_result *res,
                                double *history) {
  cg_state s;
  memset(&s, 0, sizeof s);
  s.b.apply = apply, s.b.op = op, s.b.n = n, s.b.history = history;
  take_preconditioner(&s.b);
  const solver_vt vt = {cg_init, cg_iterate, NULL};
  iterative_solve(&s.b, &vt, &s, x, b, p, res);
  free(s.p), free(s.r), free(s.z);
}

/* ---- BiCGStab: Solvers/SolverBiCgStab.hpp:52-167 ------------------------ */
typedef struct bicg_state {
  solver_base b;
  double alpha, rho, omega;
  double *p, *r, *rt, *t, *v, *z;
} bicg_state;

/* SolverBiCgStab.hpp:59-91 */
static double bicg_init(void *sv, const double *x, const double *b) {
  bicg_state *s = (bicg_state *)sv;
  const int64_t n = s->b.n;
  s->p = new_vec(n), s->r = new_vec(n), s->rt = new_vec(n); /* :65-70 */
  s->t = new_vec(n), s->v = new_vec(n);
  s->alpha = s->omega = 0.0; /* members are uninitialised in the reference;
                                they are written before first use (:139,:159) */
  s->b.applies++;
  if (s->b.pre) s->z = new_vec(n);                          /* :70 */
  op_residual(s->b.apply, s->b.op, n, s->r, b, x);          /* :82 */
  if (LEFT_PRE(&s->b)) {                                    /* :83-86 */
    double *t_ = s->z;
    s->z = s->r, s->r = t_;
    pre_mul(&s->b, s->r, s->z);
  }
  oracle_copy(n, s->rt, s->r);                              /* :87 */
  s->rho = oracle_dot(n, s->rt, s->r);                      /* :88 */
  return sqrt(s->rho);                                      /* :90 */
}
/* SolverBiCgStab.hpp:93-165 */
static double bicg_iterate(void *sv, double *x, const double *b) {
  (void)b;
  bicg_state *s = (bicg_state *)sv;
  const int64_t n = s->b.n;
  const int first_iteration = s->b.iteration == 0;                    /* :112 */
  if (first_iteration) {
    oracle_copy(n, s->p, s->r);                                       /* :114 */
  } else {
    const double rho_bar = s->rho;                                    /* :116-117 */
    s->rho = oracle_dot(n, s->rt, s->r);
    const double beta =
        oracle_safe_divide(s->alpha * s->rho, s->omega * rho_bar);    /* :118 */
    oracle_bicg_p(n, s->p, s->r, beta, s->omega, s->v);               /* :119 */
  }
  const int right_pre = RIGHT_PRE(&s->b);
  side_mul(&s->b, s->v, s->z, s->p);                                  /* :134-137 */
  s->alpha = oracle_safe_divide(s->rho, oracle_dot(n, s->rt, s->v));  /* :139 */
  oracle_axpy(n, x, s->alpha, right_pre ? s->z : s->p);               /* :140 */
  oracle_axmy(n, s->r, s->alpha, s->v);                               /* :141 */
  side_mul(&s->b, s->t, s->z, s->r);                                  /* :155-158 */
  s->omega = oracle_safe_divide(oracle_dot(n, s->t, s->r),
                                oracle_dot(n, s->t, s->t));           /* :159-160 */
  oracle_axpy(n, x, s->omega, right_pre ? s->z : s->r);               /* :161 */
  oracle_axmy(n, s->r, s->omega, s->t);                               /* :162 */
  return oracle_norm2(n, s->r);                                       /* :164 */
}

ORACLE_API void oracle_solve_bicgstab(oracle_apply_fn apply, void *op,
                                      int64_t n, double *x, const double *b,
                                      const oracle_params *p,
                                      oracle_result *res, double *history) {
  bicg_state s;
  memset(&s, 0, sizeof s);
  s.b.apply = apply, s.b.op = op, s.b.n = n, s.b.history = history;
  take_preconditioner(&s.b);
  const solver_vt vt = {bicg_init, bicg_iterate, NULL};
  iterative_solve(&s.b, &vt, &s, x, b, p, res);
  free(s.p), free(s.r), free(s.rt), free(s.t), free(s.v), free(s.z);
}

/* ---- GMRES(m) / FGMRES(m): Solvers/SolverGmres.hpp:41-255,281-308 under    */
/* InnerOuterIterativeSolver, Solver.hpp:154-259.  `pre == NULL` is the       */
/* unpreconditioned hot path; otherwise `side` picks the left (0) or right    */
/* (1) branches and `flexible` the FGMRES ones (always right, :98-99).  ----- */
typedef struct gmres_state {
  solver_base b;
  int64_t m;               /* num_inner_iterations */
  int64_t inner_iteration; /* Solver.hpp:158 */
  double *beta, *cs, *sn;  /* [m+1], [m], [m]          SolverGmres.hpp:45 */
  double *H;               /* (m+1) x m, row-major     SolverGmres.hpp:46 */
  double **q;              /* m+1 basis vectors        SolverGmres.hpp:47 */
  oracle_apply_fn pre;     /* Preconditioner::mul, or NULL */
  void *pre_ctx;
  int side, flexible;
  int64_t nz;              /* m if flexible, else 1    SolverGmres.hpp:48-49,63 */
  double **z;
  int64_t pre_applies;
} gmres_state;

static int gmres_left_pre(const gmres_state *s) { return s->pre && !s->flexible && s->side == 0; }
static int gmres_right_pre(const gmres_state *s) { return s->pre && (s->flexible || s->side == 1); }
static void gmres_pre_mul(gmres_state *s, double *y, const double *x) {
  s->pre(s->pre_ctx, y, x);
  s->pre_applies++;
}

#define H_(s, i, j) ((s)->H[(i) * (s)->m + (j)])

/* outer_init :51-91 and inner_init :93-117 share this body. */
static void gmres_start(gmres_state *s, const double *x, const double *b) {
  const int64_t n = s->b.n;
  s->b.applies++;
  op_residual(s->b.apply, s->b.op, n, s->q[0], b, x); /* :82 / :110 */
  if (gmres_left_pre(s)) {                            /* :83-86 / :111-114 */
    double *t = s->z[0];
    s->z[0] = s->q[0], s->q[0] = t;
    gmres_pre_mul(s, s->q[0], s->z[0]);
  }
  s->beta[0] = oracle_norm2(n, s->q[0]);              /* :87 / :115 */
  oracle_div_scalar(n, s->q[0], s->beta[0]);          /* :88 / :116 */
}

static double gmres_outer_init(void *sv, const double *x, const double *b) {
  gmres_state *s = (gmres_state *)sv;
  const int64_t m = s->m, n = s->b.n;
  s->beta = (double *)calloc((size_t)m + 1, sizeof(double)); /* :56-58 */
  s->cs = (double *)calloc((size_t)m, sizeof(double));
  s->sn = (double *)calloc((size_t)m, sizeof(double));
  s->H = (double *)calloc((size_t)((m + 1) * m), sizeof(double));
  s->q = (double **)calloc((size_t)m + 1, sizeof(double *)); /* :60-61 */
  for (int64_t i = 0; i <= m; ++i) s->q[i] = new_vec(n);
  if (s->pre) {                                              /* :62-65 */
    s->nz = s->flexible ? m : 1;
    s->z = (double **)calloc((size_t)s->nz, sizeof(double *));
    for (int64_t i = 0; i < s->nz; ++i) s->z[i] = new_vec(n);
  }
  gmres_start(s, x, b);
  return s->beta[0]; /* :90 */
}

/* inner_iterate :119-192 */
static double gmres_inner_iterate(gmres_state *s) {
  const int64_t n = s->b.n, k = s->inner_iteration;
  if (gmres_left_pre(s)) {                                 /* :148-149, chained mul Operator.hpp:82-88 */
    op_mul(&s->b, s->z[0], s->q[k]);
    gmres_pre_mul(s, s->q[k + 1], s->z[0]);
  } else if (gmres_right_pre(s)) {                         /* :150-152 */
    const int64_t j = s->flexible ? k : 0;
    gmres_pre_mul(s, s->z[j], s->q[k]);
    op_mul(&s->b, s->q[k + 1], s->z[j]);
  } else {
    op_mul(&s->b, s->q[k + 1], s->q[k]);                   /* :155 */
  }
  for (int64_t i = 0; i <= k; ++i) {                       /* :157-160 MGS */
    H_(s, i, k) = oracle_dot(n, s->q[k + 1], s->q[i]);
    oracle_axmy(n, s->q[k + 1], H_(s, i, k), s->q[i]);
  }
  H_(s, k + 1, k) = oracle_norm2(n, s->q[k + 1]);          /* :161 */
  oracle_div_scalar(n, s->q[k + 1], H_(s, k + 1, k));      /* :162 */
  for (int64_t i = 0; i < k; ++i) {                        /* :176-180 */
    const double chi = s->cs[i] * H_(s, i, k) + s->sn[i] * H_(s, i + 1, k);
    H_(s, i + 1, k) = -s->sn[i] * H_(s, i, k) + s->cs[i] * H_(s, i + 1, k);
    H_(s, i, k) = chi;
  }
  double rr;
  oracle_sym_ortho(H_(s, k, k), H_(s, k + 1, k), &s->cs[k], &s->sn[k], &rr); /* :181 */
  H_(s, k, k) = s->cs[k] * H_(s, k, k) + s->sn[k] * H_(s, k + 1, k);  /* :182 */
  H_(s, k + 1, k) = 0.0;                                              /* :183 */
  s->beta[k + 1] = -s->sn[k] * s->beta[k];                            /* :189 */
  s->beta[k] *= s->cs[k];
  return fabs(s->beta[k + 1]);                                        /* :191 */
}

/* inner_finalize :194-249 (not right-preconditioned branch :233-236) */
static void gmres_inner_finalize(gmres_state *s, double *x) {
  const int64_t n = s->b.n, k = s->inner_iteration;
  for (int64_t i = k; i >= 0; --i) {                        /* :207-212 */
    for (int64_t j = i + 1; j <= k; ++j) s->beta[i] -= H_(s, i, j) * s->beta[j];
    s->beta[i] /= H_(s, i, i);
  }
  if (!gmres_right_pre(s)) {
    for (int64_t i = 0; i <= k; ++i) oracle_axpy(n, x, s->beta[i], s->q[i]); /* :234-236 */
  } else if (s->flexible) {
    for (int64_t i = 0; i <= k; ++i) oracle_axpy(n, x, s->beta[i], s->z[i]); /* :238-240 */
  } else {                                                                   /* :242-247 */
    oracle_mul_scalar(n, s->q[0], s->beta[0]);
    for (int64_t i = 1; i <= k; ++i) oracle_axpy(n, s->q[0], s->beta[i], s->q[i]);
    gmres_pre_mul(s, s->z[0], s->q[0]);
    oracle_axpy(n, x, 1.0, s->z[0]);
  }
}

/* InnerOuterIterativeSolver::iterate  Solver.hpp:236-248 */
static double gmres_iterate(void *sv, double *x, const double *b) {
  gmres_state *s = (gmres_state *)sv;
  s->inner_iteration = s->b.iteration % s->m;               /* :239 */
  if (s->inner_iteration == 0) {                            /* :240-242 */
    /* NB: on the very first iteration this recomputes what outer_init just
       did (the reference has the same duplication, SolverGmres.hpp:66-67). */
    gmres_start(s, x, b);
  }
  const double rn = gmres_inner_iterate(s);                 /* :243 */
  if (s->inner_iteration == s->m - 1) gmres_inner_finalize(s, x); /* :244-246 */
  return rn;
}
/* InnerOuterIterativeSolver::finalize  Solver.hpp:250-257 */
static void gmres_finalize(void *sv, double *x, const double *b) {
  (void)b;
  gmres_state *s = (gmres_state *)sv;
  if (s->inner_iteration != s->m - 1) gmres_inner_finalize(s, x);
}

/* side: 0 = Left, 1 = Right (PreconditionerSide, Preconditioner.hpp:39-60; Symmetric takes neither
   branch in GMRES, i.e. runs unpreconditioned except for the allocation).  Returns the number of
   preconditioner applications. */
ORACLE_API int64_t oracle_solve_gmres_pre(oracle_apply_fn apply, void *op, oracle_apply_fn pre,
                                          void *pre_ctx, int side, int flexible, int64_t n,
                                          double *x, const double *b, const oracle_params *p,
                                          oracle_result *res, double *history) {
  gmres_state s;
  memset(&s, 0, sizeof s);
  s.b.apply = apply, s.b.op = op, s.b.n = n, s.b.history = history;
  take_preconditioner(&s.b);
  s.m = p->num_inner_iterations;
  if (pre == NULL && s.b.pre != NULL) pre = s.b.pre, pre_ctx = s.b.pre_ctx, side = s.b.side;  /* staged by oracle_set_preconditioner */
  s.pre = pre, s.pre_ctx = pre_ctx, s.side = side, s.flexible = flexible;
  const solver_vt vt = {gmres_outer_init, gmres_iterate, gmres_finalize};
  iterative_solve(&s.b, &vt, &s, x, b, p, res);
  for (int64_t i = 0; i <= s.m; ++i) free(s.q[i]);
  for (int64_t i = 0; i < s.nz; ++i) free(s.z[i]);
  free(s.q), free(s.z), free(s.beta), free(s.cs), free(s.sn), free(s.H);
  g_last_pre_applies = s.pre_applies;
  return s.pre_applies;
}

ORACLE_API void oracle_solve_gmres(oracle_apply_fn apply, void *op, int64_t n,
                                   double *x, const double *b,
                                   const oracle_params *p, oracle_result *res,
                                   double *history) {
  (void)oracle_solve_gmres_pre(apply, op, NULL, NULL, 1, 0, n, x, b, p, res, history);
}

/* Diagonal preconditioner y = d .* x (a Jacobi P = diag(A)^-1 passes d = 1 / diag). */
typedef struct oracle_diag_op {
  int64_t n;
  const double *d;
} oracle_diag_op;

ORACLE_API void oracle_diag_apply(void *ctx, double *y, const double *x) {
  const oracle_diag_op *op = (const oracle_diag_op *)ctx;
  for (int64_t i = 0; i < op->n; ++i) y[i] = op->d[i] * x[i];
}

/* ---- Richardson: Solvers/SolverRichardson.hpp:41-98 (ignores pre_side) --- */
typedef struct rich_state {
  solver_base b;
  double omega;
  double *r, *z;
} rich_state;

static void rich_precondition(rich_state *s) {              /* :66-69 == :90-93 */
  if (s->b.pre) {
    double *t_ = s->z;
    s->z = s->r, s->r = t_;
    pre_mul(&s->b, s->r, s->z);
  }
}
static double rich_init(void *sv, const double *x, const double *b) {
  rich_state *s = (rich_state *)sv;
  s->r = new_vec(s->b.n);                                   /* :53 */
  if (s->b.pre) s->z = new_vec(s->b.n);                     /* :54 */
  s->b.applies++;
  op_residual(s->b.apply, s->b.op, s->b.n, s->r, b, x);     /* :65 */
  rich_precondition(s);
  return oracle_norm2(s->b.n, s->r);                        /* :71 */
}
static double rich_iterate(void *sv, double *x, const double *b) {
  rich_state *s = (rich_state *)sv;
  oracle_axpy(s->b.n, x, s->omega, s->r);                   /* :88  x += omega r */
  s->b.applies++;
  op_residual(s->b.apply, s->b.op, s->b.n, s->r, b, x);     /* :89 */
  rich_precondition(s);
  return oracle_norm2(s->b.n, s->r);                        /* :95 */
}
ORACLE_API void oracle_solve_richardson(oracle_apply_fn apply, void *op, int64_t n, double *x,
                                        const double *b, const oracle_params *p, oracle_result *res,
                                        double *history) {
  rich_state s;
  memset(&s, 0, sizeof s);
  s.b.apply = apply, s.b.op = op, s.b.n = n, s.b.history = history;
  take_preconditioner(&s.b);
  s.omega = p->relaxation_factor;
  const solver_vt vt = {rich_init, rich_iterate, NULL};
  iterative_solve(&s.b, &vt, &s, x, b, p, res);
  free(s.r), free(s.z);
}

/* ---- CGS: Solvers/SolverCgs.hpp:50-176 ---------------------------------- */
typedef struct cgs_state {
  solver_base b;
  double rho;
  double *p, *q, *r, *rt, *u, *v;
} cgs_state;

static double cgs_init(void *sv, const double *x, const double *b) {
  cgs_state *s = (cgs_state *)sv;
  const int64_t n = s->b.n;
  s->p = new_vec(n), s->q = new_vec(n), s->r = new_vec(n);  /* :62-67 */
  s->rt = new_vec(n), s->u = new_vec(n), s->v = new_vec(n);
  s->b.applies++;
  op_residual(s->b.apply, s->b.op, n, s->r, b, x);          /* :80 */
  if (LEFT_PRE(&s->b)) {                                    /* :81-84 */
    double *t_ = s->u;
    s->u = s->r, s->r = t_;
    pre_mul(&s->b, s->r, s->u);
  }
  oracle_copy(n, s->rt, s->r);                              /* :85 */
  s->rho = oracle_dot(n, s->rt, s->r);                      /* :86 */
  return sqrt(s->rho);                                      /* :88 */
}
static double cgs_iterate(void *sv, double *x, const double *b) {
  (void)b;
  cgs_state *s = (cgs_state *)sv;
  const int64_t n = s->b.n;
  if (s->b.iteration == 0) {                                /* :113-116 */
    oracle_copy(n, s->u, s->r);
    oracle_copy(n, s->p, s->u);
  } else {
    const double rho_bar = s->rho;                          /* :118-119 */
    s->rho = oracle_dot(n, s->rt, s->r);
    const double beta = oracle_safe_divide(s->rho, rho_bar); /* :120 */
    for (int64_t i = 0; i < n; ++i) s->u[i] = s->r[i] + beta * s->q[i];                 /* :121 */
    for (int64_t i = 0; i < n; ++i) s->p[i] = s->u[i] + beta * (s->q[i] + beta * s->p[i]); /* :122 */
  }
  side_mul(&s->b, s->v, s->q, s->p);                        /* :137-139 */
  const double alpha = oracle_safe_divide(s->rho, oracle_dot(n, s->rt, s->v)); /* :140 */
  for (int64_t i = 0; i < n; ++i) s->q[i] = s->u[i] - alpha * s->v[i];  /* :141 */
  for (int64_t i = 0; i < n; ++i) s->v[i] = s->u[i] + s->q[i];          /* :142 */
  if (LEFT_PRE(&s->b)) {                                    /* :159-162 */
    oracle_axpy(n, x, alpha, s->v);
    op_mul(&s->b, s->u, s->v);                              /* pre_op->mul(v, u, lin_op, v) */
    pre_mul(&s->b, s->v, s->u);
    oracle_axmy(n, s->r, alpha, s->v);
  } else if (RIGHT_PRE(&s->b)) {                            /* :163-166 */
    pre_mul(&s->b, s->u, s->v);                             /* lin_op.mul(v, u, *pre_op, v) */
    op_mul(&s->b, s->v, s->u);
    oracle_axpy(n, x, alpha, s->u);
    oracle_axmy(n, s->r, alpha, s->v);
  } else {
    op_mul(&s->b, s->u, s->v);                              /* :167 */
    oracle_axpy(n, x, alpha, s->v);                         /* :168 */
    oracle_axmy(n, s->r, alpha, s->u);                      /* :169 */
  }
  return oracle_norm2(n, s->r);                             /* :172 */
}
ORACLE_API void oracle_solve_cgs(oracle_apply_fn apply, void *op, int64_t n, double *x, const double *b,
                                 const oracle_params *p, oracle_result *res, double *history) {
  cgs_state s;
  memset(&s, 0, sizeof s);
  s.b.apply = apply, s.b.op = op, s.b.n = n, s.b.history = history;
  take_preconditioner(&s.b);
  const solver_vt vt = {cgs_init, cgs_iterate, NULL};
  iterative_solve(&s.b, &vt, &s, x, b, p, res);
  free(s.p), free(s.q), free(s.r), free(s.rt), free(s.u), free(s.v);
}

/* ---- TFQMR / TFQMR1: Solvers/SolverTfqmr.hpp:37-265 --------------------- */
typedef struct tfqmr_state {
  solver_base b;
  int l1;
  double rho, tau;
  double *d, *rt, *u, *v, *y, *s, *z;
} tfqmr_state;

static double tfqmr_init(void *sv, const double *x, const double *b) {
  tfqmr_state *s = (tfqmr_state *)sv;
  const int64_t n = s->b.n;
  s->d = new_vec(n), s->rt = new_vec(n), s->u = new_vec(n);  /* :49-55 */
  s->v = new_vec(n), s->y = new_vec(n), s->s = new_vec(n);
  if (s->l1) oracle_copy(n, s->d, x);                       /* :73-77 */
  else oracle_fill(n, s->d, 0.0);
  s->b.applies++;
  if (s->b.pre) s->z = new_vec(n);                          /* :56 */
  op_residual(s->b.apply, s->b.op, n, s->y, b, x);          /* :78 */
  if (LEFT_PRE(&s->b)) {                                    /* :79-82 */
    double *t_ = s->z;
    s->z = s->y, s->y = t_;
    pre_mul(&s->b, s->y, s->z);
  }
  oracle_copy(n, s->u, s->y);                               /* :83 */
  oracle_copy(n, s->rt, s->u);                              /* :84 */
  s->rho = oracle_dot(n, s->rt, s->u);                      /* :85 */
  s->tau = sqrt(s->rho);
  return s->tau;                                            /* :87 */
}
static double tfqmr_iterate(void *sv, double *x, const double *b) {
  (void)b;
  tfqmr_state *s = (tfqmr_state *)sv;
  const int64_t n = s->b.n;
  const int right_pre = RIGHT_PRE(&s->b);
  if (s->b.iteration == 0) {                                /* :121-126 */
    side_mul(&s->b, s->s, s->z, s->y);
    oracle_copy(n, s->v, s->s);
  } else {
    const double rho_bar = s->rho;                          /* :128-129 */
    s->rho = oracle_dot(n, s->rt, s->u);
    const double beta = oracle_safe_divide(s->rho, rho_bar); /* :130 */
    oracle_xpay(n, s->v, s->s, beta);                       /* :131  v <<= s + beta v */
    oracle_xpay(n, s->y, s->u, beta);                       /* :132  y <<= u + beta y */
    side_mul(&s->b, s->s, s->z, s->y);                      /* :133-135 */
    oracle_xpay(n, s->v, s->s, beta);                       /* :136 */
  }
  const double alpha = oracle_safe_divide(s->rho, oracle_dot(n, s->rt, s->v)); /* :166 */
  for (int m = 0; m <= 1; ++m) {                            /* :167-189 */
    oracle_axmy(n, s->u, alpha, s->s);                      /* :168 */
    oracle_axpy(n, s->d, alpha, right_pre ? s->z : s->y);   /* :169 */
    const double omega = oracle_norm2(n, s->u);             /* :170 */
    if (s->l1) {
      if (omega < s->tau) s->tau = omega, oracle_copy(n, x, s->d); /* :172 */
    } else {
      double cs, sn, rr;
      oracle_sym_ortho(s->tau, omega, &cs, &sn, &rr);       /* :174 */
      s->tau = omega * cs;                                  /* :175 */
      oracle_axpy(n, x, pow(cs, 2), s->d);                  /* :176 */
      oracle_mul_scalar(n, s->d, pow(sn, 2));               /* :177 */
    }
    if (m == 0) {
      oracle_axmy(n, s->y, alpha, s->v);                    /* :180 */
      side_mul(&s->b, s->s, s->z, s->y);                    /* :181-183 */
    }
  }
  double tau_tilde = s->tau;                                /* :199-204 */
  if (!s->l1) tau_tilde *= sqrt(2.0 * (double)s->b.iteration + 3.0);
  return tau_tilde;
}
static void tfqmr_solve(int l1, oracle_apply_fn apply, void *op, int64_t n, double *x, const double *b,
                        const oracle_params *p, oracle_result *res, double *history) {
  tfqmr_state s;
  memset(&s, 0, sizeof s);
  s.b.apply = apply, s.b.op = op, s.b.n = n, s.b.history = history;
  take_preconditioner(&s.b);
  s.l1 = l1;
  const solver_vt vt = {tfqmr_init, tfqmr_iterate, NULL};
  iterative_solve(&s.b, &vt, &s, x, b, p, res);
  free(s.d), free(s.rt), free(s.u), free(s.v), free(s.y), free(s.s), free(s.z);
}
ORACLE_API void oracle_solve_tfqmr(oracle_apply_fn apply, void *op, int64_t n, double *x, const double *b,
                                   const oracle_params *p, oracle_result *res, double *history) {
  tfqmr_solve(0, apply, op, n, x, b, p, res, history);
}
ORACLE_API void oracle_solve_tfqmr1(oracle_apply_fn apply, void *op, int64_t n, double *x, const double *b,
                                    const oracle_params *p, oracle_result *res, double *history) {
  tfqmr_solve(1, apply, op, n, x, b, p, res, history);
}

/* ---- fill_randomly: Bittern/MatrixAlgorithms.hpp:140-153 -------------------
 * A function-static std::mt19937_64{} (default seed 5489; state persists across calls) feeding
 * std::uniform_real_distribution(0, 1).  MT19937-64 is the published Matsumoto-Nishimura
 * generator (the C++ standard fixes its parameters and its 10000th output,
 * 9981545732273789042); libstdc++'s distribution draws ONE 64-bit word per double and returns
 * double(word) / 2^64, clipped below 1 (bits/random.tcc, generate_canonical). */
#define MT_NN 312
#define MT_MM 156
static uint64_t mt_state[MT_NN];
static int mt_index = MT_NN + 1;

ORACLE_API void oracle_rng_reset(void) {
  mt_state[0] = 5489ULL;
  for (int i = 1; i < MT_NN; ++i)
    mt_state[i] = 6364136223846793005ULL * (mt_state[i - 1] ^ (mt_state[i - 1] >> 62)) + (uint64_t)i;
  mt_index = MT_NN;
}
ORACLE_API uint64_t oracle_rng_next(void) {
  if (mt_index > MT_NN) oracle_rng_reset();
  if (mt_index == MT_NN) {
    for (int i = 0; i < MT_NN; ++i) {
      const uint64_t x = (mt_state[i] & 0xFFFFFFFF80000000ULL) | (mt_state[(i + 1) % MT_NN] & 0x7FFFFFFFULL);
      mt_state[i] = mt_state[(i + MT_MM) % MT_NN] ^ (x >> 1) ^ ((x & 1ULL) ? 0xB5026F5AA96619E9ULL : 0ULL);
    }
    mt_index = 0;
  }
  uint64_t x = mt_state[mt_index++];
  x ^= (x >> 29) & 0x5555555555555555ULL;
  x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
  x ^= (x << 37) & 0xFFF7EEE000000000ULL;
  x ^= (x >> 43);
  return x;
}
ORACLE_API void oracle_fill_randomly(int64_t n, double *y) {
  for (int64_t i = 0; i < n; ++i) {
    double r = (double)oracle_rng_next() / 18446744073709551616.0;
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    y[i] = r;
  }
}

/* ---- BiCGStab(l): Solvers/SolverBiCgStab.hpp:184-383 (a preconditioner is  */
/* always applied on the left, whatever pre_side says) ---------------------- */
typedef struct bicgl_state {
  solver_base b;
  int64_t l;
  double alpha, rho, omega;
  double *gamma, *gamma_bar, *gamma_bbar, *sigma, *tau; /* tau is (l+1) x (l+1) */
  double *rt, *z;
  double **r, **u;
} bicgl_state;
#define TAU_(s, i, j) ((s)->tau[(i) * ((s)->l + 1) + (j)])

static double bicgl_init(void *sv, const double *x, const double *b) {
  bicgl_state *s = (bicgl_state *)sv;
  const int64_t n = s->b.n, l = s->l;
  s->gamma = (double *)calloc((size_t)l + 1, sizeof(double));       /* :202-206 */
  s->gamma_bar = (double *)calloc((size_t)l + 1, sizeof(double));
  s->gamma_bbar = (double *)calloc((size_t)l + 1, sizeof(double));
  s->sigma = (double *)calloc((size_t)l + 1, sizeof(double));
  s->tau = (double *)calloc((size_t)((l + 1) * (l + 1)), sizeof(double));
  s->rt = new_vec(n);
  s->r = (double **)calloc((size_t)l + 1, sizeof(double *));
  s->u = (double **)calloc((size_t)l + 1, sizeof(double *));
  for (int64_t i = 0; i <= l; ++i) s->r[i] = new_vec(n), s->u[i] = new_vec(n);
  oracle_fill(n, s->u[0], 0.0);                                      /* :224 */
  s->b.applies++;
  if (s->b.pre) s->z = new_vec(n);                                   /* :217 */
  op_residual(s->b.apply, s->b.op, n, s->r[0], b, x);                /* :225 */
  if (s->b.pre) {                                                    /* :226-229 */
    double *t_ = s->z;
    s->z = s->r[0], s->r[0] = t_;
    pre_mul(&s->b, s->r[0], s->z);
  }
  oracle_copy(n, s->rt, s->r[0]);                                    /* :230 */
  s->rho = oracle_dot(n, s->rt, s->r[0]);                            /* :231 */
  return sqrt(s->rho);
}
static double bicgl_iterate(void *sv, double *x, const double *b) {
  (void)b;
  bicgl_state *s = (bicgl_state *)sv;
  const int64_t n = s->b.n, l = s->l;
  const int64_t j = s->b.iteration % l;                              /* Solver.hpp:239 */
  if (s->b.iteration == 0) {
    oracle_copy(n, s->u[0], s->r[0]);                                /* :264 */
  } else {
    const double rho_bar = s->rho;                                   /* :266-267 */
    s->rho = oracle_dot(n, s->rt, s->r[j]);
    const double beta = oracle_safe_divide(s->alpha * s->rho, rho_bar); /* :268 */
    for (int64_t i = 0; i <= j; ++i)                                 /* :269-271 */
      for (int64_t q = 0; q < n; ++q) s->u[i][q] = s->r[i][q] - beta * s->u[i][q];
  }
  if (s->b.pre) {                                                    /* :273-274 */
    op_mul(&s->b, s->z, s->u[j]);
    pre_mul(&s->b, s->u[j + 1], s->z);
  } else {
    op_mul(&s->b, s->u[j + 1], s->u[j]);                             /* :276 */
  }
  s->alpha = oracle_safe_divide(s->rho, oracle_dot(n, s->rt, s->u[j + 1])); /* :278 */
  for (int64_t i = 0; i <= j; ++i) oracle_axmy(n, s->r[i], s->alpha, s->u[i + 1]); /* :279-281 */
  oracle_axpy(n, x, s->alpha, s->u[0]);                              /* :291 */
  if (s->b.pre) {                                                    /* :292-293 */
    op_mul(&s->b, s->z, s->r[j]);
    pre_mul(&s->b, s->r[j + 1], s->z);
  } else {
    op_mul(&s->b, s->r[j + 1], s->r[j]);                             /* :295 */
  }
  if (j == l - 1) {                                                  /* :298-365 */
    for (int64_t jj = 1; jj <= l; ++jj) {
      for (int64_t i = 1; i < jj; ++i) {
        TAU_(s, i, jj) = oracle_safe_divide(oracle_dot(n, s->r[i], s->r[jj]), s->sigma[i]);
        oracle_axmy(n, s->r[jj], TAU_(s, i, jj), s->r[i]);
      }
      s->sigma[jj] = oracle_dot(n, s->r[jj], s->r[jj]);
      s->gamma_bar[jj] = oracle_safe_divide(oracle_dot(n, s->r[0], s->r[jj]), s->sigma[jj]);
    }
    s->omega = s->gamma[l] = s->gamma_bar[l];
    s->rho *= -s->omega;
    for (int64_t jj = l - 1; jj != 0; --jj) {
      s->gamma[jj] = s->gamma_bar[jj];
      for (int64_t i = jj + 1; i <= l; ++i) s->gamma[jj] -= TAU_(s, jj, i) * s->gamma[i];
    }
    for (int64_t jj = 1; jj < l; ++jj) {
      s->gamma_bbar[jj] = s->gamma[jj + 1];
      for (int64_t i = jj + 1; i < l; ++i) s->gamma_bbar[jj] += TAU_(s, jj, i) * s->gamma[i + 1];
    }
    oracle_axpy(n, x, s->gamma[1], s->r[0]);
    oracle_axmy(n, s->r[0], s->gamma_bar[l], s->r[l]);
    oracle_axmy(n, s->u[0], s->gamma[l], s->u[l]);
    for (int64_t jj = 1; jj < l; ++jj) {
      oracle_axpy(n, x, s->gamma_bbar[jj], s->r[jj]);
      oracle_axmy(n, s->r[0], s->gamma_bar[jj], s->r[jj]);
      oracle_axmy(n, s->u[0], s->gamma[jj], s->u[jj]);
    }
  }
  return oracle_norm2(n, s->r[0]);                                   /* :367 */
}
ORACLE_API void oracle_solve_bicgstabl(oracle_apply_fn apply, void *op, int64_t n, double *x, const double *b,
                                       const oracle_params *p, oracle_result *res, double *history) {
  bicgl_state s;
  memset(&s, 0, sizeof s);
  s.b.apply = apply, s.b.op = op, s.b.n = n, s.b.history = history;
  take_preconditioner(&s.b);
  s.l = p->num_inner_iterations;  /* default 2, :379-381 */
  const solver_vt vt = {bicgl_init, bicgl_iterate, NULL};
  iterative_solve(&s.b, &vt, &s, x, b, p, res);
  for (int64_t i = 0; i <= s.l; ++i) free(s.r[i]), free(s.u[i]);
  free(s.r), free(s.u), free(s.rt), free(s.z), free(s.gamma), free(s.gamma_bar), free(s.gamma_bbar), free(s.sigma), free(s.tau);
}

/* ---- IDR(s): Solvers/SolverIdrs.hpp:52-291 ------------------------------- */
typedef struct idrs_state {
  solver_base b;
  int64_t s;
  double omega;
  double *phi, *gamma, *mu; /* mu is s x s */
  double *r, *v, *z;
  double **p, **u, **g;
} idrs_state;
#define MU_(st, i, j) ((st)->mu[(i) * (st)->s + (j)])

static double idrs_init(void *sv, const double *x, const double *b) {
  idrs_state *st = (idrs_state *)sv;
  const int64_t n = st->b.n, s = st->s;
  st->phi = (double *)calloc((size_t)s, sizeof(double));            /* :73-75 */
  st->gamma = (double *)calloc((size_t)s, sizeof(double));
  st->mu = (double *)calloc((size_t)(s * s), sizeof(double));
  st->r = new_vec(n), st->v = new_vec(n);
  st->p = (double **)calloc((size_t)s, sizeof(double *));
  st->u = (double **)calloc((size_t)s, sizeof(double *));
  st->g = (double **)calloc((size_t)s, sizeof(double *));
  for (int64_t i = 0; i < s; ++i) st->p[i] = new_vec(n), st->u[i] = new_vec(n), st->g[i] = new_vec(n);
  st->b.applies++;
  if (st->b.pre) st->z = new_vec(n);                                 /* :79 */
  op_residual(st->b.apply, st->b.op, n, st->r, b, x);               /* :99 */
  if (LEFT_PRE(&st->b)) {                                            /* :100-103 */
    double *t_ = st->z;
    st->z = st->r, st->r = t_;
    pre_mul(&st->b, st->r, st->z);
  }
  st->phi[0] = oracle_norm2(n, st->r);                               /* :104 */
  return st->phi[0];
}
static void idrs_inner_init(idrs_state *st) {                        /* :109-156 */
  const int64_t n = st->b.n, s = st->s;
  if (st->b.iteration == 0) {
    st->omega = MU_(st, 0, 0) = 1.0;
    for (int64_t q = 0; q < n; ++q) st->p[0][q] = st->r[q] / st->phi[0]; /* :131 */
    for (int64_t i = 1; i < s; ++i) {
      MU_(st, i, i) = 1.0, st->phi[i] = 0.0;
      oracle_fill_randomly(n, st->p[i]);                             /* :135 */
      for (int64_t j = 0; j < i; ++j) {
        MU_(st, i, j) = 0.0;
        oracle_axmy(n, st->p[i], oracle_dot(n, st->p[i], st->p[j]), st->p[j]); /* :138 */
      }
      oracle_div_scalar(n, st->p[i], oracle_norm2(n, st->p[i]));     /* :140 */
    }
  } else {
    for (int64_t i = 0; i < s; ++i) st->phi[i] = oracle_dot(n, st->p[i], st->r); /* :143-145 */
  }
}
static double idrs_iterate(void *sv, double *x, const double *b) {
  (void)b;
  idrs_state *st = (idrs_state *)sv;
  const int64_t n = st->b.n, s = st->s;
  const int64_t k = st->b.iteration % s;                             /* Solver.hpp:239 */
  if (k == 0) idrs_inner_init(st);                                   /* Solver.hpp:240-242 */
  for (int64_t i = k; i < s; ++i) {                                  /* :182-188 */
    st->gamma[i] = st->phi[i];
    for (int64_t j = k; j < i; ++j) st->gamma[i] -= MU_(st, i, j) * st->gamma[j];
    st->gamma[i] /= MU_(st, i, i);
  }
  for (int64_t q = 0; q < n; ++q) st->v[q] = st->r[q] - st->gamma[k] * st->g[k][q];   /* :200 */
  for (int64_t i = k + 1; i < s; ++i) oracle_axmy(n, st->v, st->gamma[i], st->g[i]); /* :201-203 */
  if (RIGHT_PRE(&st->b)) {                                           /* :204-207 */
    double *t_ = st->z;
    st->z = st->v, st->v = t_;
    pre_mul(&st->b, st->v, st->z);
  }
  for (int64_t q = 0; q < n; ++q) st->u[k][q] = st->omega * st->v[q] + st->gamma[k] * st->u[k][q]; /* :208 */
  for (int64_t i = k + 1; i < s; ++i) oracle_axpy(n, st->u[k], st->gamma[i], st->u[i]);           /* :209-211 */
  if (LEFT_PRE(&st->b)) {                                            /* :212-213 */
    op_mul(&st->b, st->z, st->u[k]);
    pre_mul(&st->b, st->g[k], st->z);
  } else {
    op_mul(&st->b, st->g[k], st->u[k]);                              /* :215 */
  }
  for (int64_t i = 0; i < k; ++i) {                                  /* :230-235 */
    const double alpha = oracle_safe_divide(oracle_dot(n, st->p[i], st->g[k]), MU_(st, i, i));
    oracle_axmy(n, st->u[k], alpha, st->u[i]);
    oracle_axmy(n, st->g[k], alpha, st->g[i]);
  }
  for (int64_t i = k; i < s; ++i) MU_(st, i, k) = oracle_dot(n, st->p[i], st->g[k]); /* :236-238 */
  const double beta = oracle_safe_divide(st->phi[k], MU_(st, k, k)); /* :250 */
  oracle_axpy(n, x, beta, st->u[k]);
  oracle_axmy(n, st->r, beta, st->g[k]);
  for (int64_t i = k + 1; i < s; ++i) st->phi[i] -= beta * MU_(st, i, k);
  if (k == s - 1) {                                                  /* :256-279 */
    side_mul(&st->b, st->v, st->z, st->r);                           /* :270-272 */
    st->omega = oracle_safe_divide(oracle_dot(n, st->v, st->r), oracle_dot(n, st->v, st->v));
    oracle_axpy(n, x, st->omega, RIGHT_PRE(&st->b) ? st->z : st->r); /* :276 */
    oracle_axmy(n, st->r, st->omega, st->v);
  }
  return oracle_norm2(n, st->r);                                     /* :281 */
}
ORACLE_API void oracle_solve_idrs(oracle_apply_fn apply, void *op, int64_t n, double *x, const double *b,
                                  const oracle_params *p, oracle_result *res, double *history) {
  idrs_state st;
  memset(&st, 0, sizeof st);
  st.b.apply = apply, st.b.op = op, st.b.n = n, st.b.history = history;
  take_preconditioner(&st.b);
  st.s = p->num_inner_iterations;  /* default 4, :287-289 */
  const solver_vt vt = {idrs_init, idrs_iterate, NULL};
  iterative_solve(&st.b, &vt, &st, x, b, p, res);
  for (int64_t i = 0; i < st.s; ++i) free(st.p[i]), free(st.u[i]), free(st.g[i]);
  free(st.p), free(st.u), free(st.g), free(st.r), free(st.v), free(st.z), free(st.phi), free(st.gamma), free(st.mu);
}

ORACLE_API int oracle_abi_version(void) { return 3; }

/* ---- JFNK: Solvers/SolverNewton.hpp:101-173.  `apply` is the (possibly     */
/* nonlinear) operator A(x); every outer iteration solves J(x) t = r with the  */
/* reference's inner BiCGStab (tolerances 1e-8, default 2000 iterations,       */
/* :133-135) where J(x) y = (A(x + delta y) - A(x)) / delta, :136-148. ------ */
typedef struct jfnk_state {
  solver_base b;
  double *s, *t, *r, *w;
  const double *x; /* current iterate, read by the Jacobian-vector product */
  double mu;
  int64_t inner_iterations;
} jfnk_state;

static void jfnk_jacobian_apply(void *ctx, double *z, const double *y) {
  jfnk_state *st = (jfnk_state *)ctx;
  const int64_t n = st->b.n;
  const double delta = oracle_safe_divide(st->mu, oracle_norm2(n, y));      /* :144 */
  for (int64_t i = 0; i < n; ++i) st->s[i] = st->x[i] + delta * y[i];       /* :145 */
  op_mul(&st->b, z, st->s);                                                 /* :146 */
  const double delta_inverse = oracle_safe_divide(1.0, delta);              /* :147 */
  for (int64_t i = 0; i < n; ++i) z[i] = delta_inverse * (z[i] - st->w[i]); /* :148 */
}

static double jfnk_init(void *sv, const double *x, const double *b) {
  jfnk_state *st = (jfnk_state *)sv;
  const int64_t n = st->b.n;
  st->s = new_vec(n), st->t = new_vec(n), st->r = new_vec(n), st->w = new_vec(n); /* :108-111 */
  op_mul(&st->b, st->w, x);                                                 /* :118 */
  for (int64_t i = 0; i < n; ++i) st->r[i] = b[i] - st->w[i];               /* :119 */
  return oracle_norm2(n, st->r);
}

static double jfnk_iterate(void *sv, double *x, const double *b) {
  jfnk_state *st = (jfnk_state *)sv;
  const int64_t n = st->b.n;
  st->mu = sqrt(2.220446049250313e-16) * sqrt(1.0 + oracle_norm2(n, x));   /* :128-130 */
  oracle_copy(n, st->t, st->r);                                             /* :131 */
  st->x = x;
  oracle_params ip = {2000, 1.0e-8, 1.0e-8, 50, 1.0e-4};                    /* :133-135 */
  oracle_result ir;
  const int64_t outer_applies = st->b.applies; /* the inner solve counts its own; fold them in */
  oracle_solve_bicgstab(jfnk_jacobian_apply, st, n, st->t, st->r, &ip, &ir, NULL);
  (void)outer_applies;
  st->inner_iterations += ir.iterations;
  oracle_axpy(n, x, 1.0, st->t);                                            /* :156 */
  op_mul(&st->b, st->w, x);                                                 /* :157 */
  for (int64_t i = 0; i < n; ++i) st->r[i] = b[i] - st->w[i];               /* :158 */
  return oracle_norm2(n, st->r);
}

/* Returns the total number of inner BiCGStab iterations. */
ORACLE_API int64_t oracle_solve_jfnk(oracle_apply_fn apply, void *op, int64_t n, double *x,
                                     const double *b, const oracle_params *p, oracle_result *res,
                                     double *history) {
  jfnk_state st;
  memset(&st, 0, sizeof st);
  st.b.apply = apply, st.b.op = op, st.b.n = n, st.b.history = history;
  take_preconditioner(&st.b);
  const solver_vt vt = {jfnk_init, jfnk_iterate, NULL};
  iterative_solve(&st.b, &vt, &st, x, b, p, res);
  free(st.s), free(st.t), free(st.r), free(st.w);
  return st.inner_iterations;
}

/* ---- The operator the reference's only caller hands to CG: the lambda of       */
/* source_apps/playground/Playground.cpp:153-167: two           */
/* stormDivGrad calls and two element loops per application.  Only the             */
/* application is restated (parity of the composite apply); the Cahn-Hilliard      */
/* time stepping around it is app physics, out of scope. ------------------------ */
typedef struct oracle_ch_op {
  const oracle_mesh *mesh;
  const double *f, *c;
  double *w_hat;
  double tau, Gamma, sigma;
} oracle_ch_op;

ORACLE_API void oracle_ch_apply(void *ctx, double *c_hat, const double *c_in) {
  const oracle_ch_op *op = (const oracle_ch_op *)ctx;
  const int64_t n = op->mesh->n_cells;
  for (int64_t i = 0; i < n; ++i) op->w_hat[i] = op->f[i] + op->sigma * (c_in[i] - op->c[i]); /* :155 */
  oracle_divgrad(op->mesh, op->w_hat, -op->Gamma, c_in);                                        /* :157 */
  for (int64_t i = 0; i < n; ++i) c_hat[i] = c_in[i];                                           /* :161 */
  oracle_divgrad(op->mesh, c_hat, -op->tau, op->w_hat);                                         /* :164 */
}

/* f <<= map(dF_dc, c), :142-148 */
ORACLE_API void oracle_ch_dF_dc(int64_t n, double *f, const double *c) {
  for (int64_t i = 0; i < n; ++i) f[i] = 2.0 * c[i] * (c[i] - 1.0) * (2.0 * c[i] - 1.0);
}
