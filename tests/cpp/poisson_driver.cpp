// A solver driver written the way StormRuler's playground writes one
// (source_apps/playground/Playground.cpp:151-167): build the mesh quantities, wrap the stencil in
// an operator, call solve<XSolver>(x, b, op).  Compiled against include/storm_hip/Storm.hpp only.
//
//   poisson_driver <n> <cg|bicgstab|gmres|fgmres|jfnk|...|user-steepest-descent|user-cg> <native|lambda|jacobi|jacobi-left|stepping|eager> [restart]
//
// prints one JSON line.  "lambda" passes the operator through make_operator (forcing the
// statement-by-statement solver templates over the BLAS-1 ABI); "native" passes a
// HipStencilOperator (whole solve on the device); "jacobi[-left]" adds the device-side diagonal
// preconditioner through the reference's pre_op / pre_side hook (Solver.hpp:74-75).
#include <storm_hip/Storm.hpp>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace Storm;

struct BoxMesh {
  std::vector<int64_t> inner, outer, b_cell;
  std::vector<real_t> coef, b_coef, volume;
  size_t n_cells = 0;
};

// n^3 unit cube, cell id (k*n + j)*n + i, faces cell-major +x,+y,+z, wall faces -x,+x,-y,+y,-z,+z
// (same synthetic mesh as stormruler_amd.mesh.structured_box / SURVEY.md 8d).
static BoxMesh make_box(int n) {
  BoxMesh m;
  const real_t h = 1.0 / n;
  m.n_cells = (size_t)n * n * n;
  m.volume.assign(m.n_cells, h * h * h);
  auto center = [&](int i) { return (i + 0.5) * h; };
  auto dist = [&](real_t a, real_t b) {  // length(a - b) of Bittern: sqrt(0 + d*d)
    const real_t d = a - b;
    real_t s = 0.0;
    s = s + d * d;
    return std::sqrt(s);
  };
  const real_t area = h * h;
  for (int k = 0; k < n; ++k)
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        const int64_t c = ((int64_t)k * n + j) * n + i;
        if (i < n - 1) m.inner.push_back(c), m.outer.push_back(c + 1), m.coef.push_back(area / dist(center(i + 1), center(i)));
        if (j < n - 1) m.inner.push_back(c), m.outer.push_back(c + n), m.coef.push_back(area / dist(center(j + 1), center(j)));
        if (k < n - 1) m.inner.push_back(c), m.outer.push_back(c + (int64_t)n * n), m.coef.push_back(area / dist(center(k + 1), center(k)));
        const int idx[3] = {i, j, k};
        for (int ax = 0; ax < 3; ++ax) {
          if (idx[ax] == 0) m.b_cell.push_back(c), m.b_coef.push_back(area / dist(center(0) - 0.5 * h, center(0)));
          if (idx[ax] == n - 1) m.b_cell.push_back(c), m.b_coef.push_back(area / dist(center(n - 1) + 0.5 * h, center(n - 1)));
        }
      }
  return m;
}

// A solver a USER of the interface writes: derive from IterativeSolver, implement the protected hooks with the vector
// statements of the overload census, and `solve` (final in the base, Solver.hpp:116-147) runs the reference's host loop
// over them.  Steepest descent: r = b - A x; alpha = <r,r> / <r,A r>; x += alpha r.
template<class Vector>
class SteepestDescentSolver final : public IterativeSolver<Vector> {
  Vector _r_vec, _z_vec;
  real_t init(const Vector& x_vec, const Vector& b_vec, const Operator<Vector>& any_op,
              const Preconditioner<Vector>*) override {
    _r_vec.assign(x_vec, false);
    _z_vec.assign(x_vec, false);
    any_op.Residual(_r_vec, b_vec, x_vec);
    return norm_2(_r_vec);
  }
  real_t iterate(Vector& x_vec, const Vector&, const Operator<Vector>& any_op, const Preconditioner<Vector>*) override {
    any_op.mul(_z_vec, _r_vec);
    const real_t alpha = safe_divide(dot_product(_r_vec, _r_vec), dot_product(_r_vec, _z_vec));
    x_vec += alpha * _r_vec;
    _r_vec -= alpha * _z_vec;
    return norm_2(_r_vec);
  }
};

// ... and the reference's CgSolver body typed against the interface, statement for statement (SolverCg.hpp:54-126): what a
// maintainer's own CG looks like.  Its host loop runs with the library's lazy statements (IterativeSolver::lazy_statements):
// `r -= alpha z; dot_product(r, r)` is ONE kernel, `mul(z, p); dot_product(p, z)` the apply with its fused dot, and on a large
// lattice `x += alpha p; p <<= r + beta p; mul(z, p); dot_product(p, z)` the library's own fused CG step (csrc/lazy.hip).
template<class Vector>
class StatementCgSolver final : public IterativeSolver<Vector> {
  real_t _gamma{0.0};
  Vector _p_vec, _r_vec, _z_vec;
  real_t init(const Vector& x_vec, const Vector& b_vec, const Operator<Vector>& any_op, const Preconditioner<Vector>*) override {
    _p_vec.assign(x_vec, false);
    _r_vec.assign(x_vec, false);
    _z_vec.assign(x_vec, false);
    any_op.Residual(_r_vec, b_vec, x_vec);
    _p_vec <<= _r_vec;
    _gamma = dot_product(_r_vec, _r_vec);
    return std::sqrt(_gamma);
  }
  real_t iterate(Vector& x_vec, const Vector&, const Operator<Vector>& any_op, const Preconditioner<Vector>*) override {
    any_op.mul(_z_vec, _p_vec);
    const real_t alpha = safe_divide(_gamma, dot_product(_p_vec, _z_vec));
    x_vec += alpha * _p_vec;
    _r_vec -= alpha * _z_vec;
    const real_t gamma_bar = _gamma;
    _gamma = dot_product(_r_vec, _r_vec);
    const real_t beta = safe_divide(_gamma, gamma_bar);
    _p_vec <<= _r_vec + beta * _p_vec;
    return std::sqrt(_gamma);
  }
};

template<template<class> class SolverT>
static int run(int n, const std::string& mode, size_t restart) {
  const bool native = mode != "lambda";
  Context ctx(0);
  const BoxMesh mesh = make_box(n);
  const StencilMatrix matrix = StencilMatrix::from_faces(ctx, mesh.n_cells, 0, mesh.inner, mesh.outer, mesh.coef,
                                                         mesh.b_cell, mesh.b_coef, mesh.volume);
  DeviceVector b(ctx, mesh.n_cells), x(ctx, mesh.n_cells);
  fill_with(b, 1.0);

  std::string logged;
  set_log_sink([&](const std::string& line) { logged = line; });  // the reference's per-solve log line
  SolverT<DeviceVector> solver;
  if constexpr (std::is_base_of_v<InnerOuterIterativeSolver<DeviceVector>, SolverT<DeviceVector>>)
    solver.num_inner_iterations = restart;
  if (mode == "stepping") solver.device_loop = false;  // the host loop over init / iterate / finalize
  if (mode == "eager") solver.lazy_statements = false;  // a host loop whose every statement is a launch when it is called
  if (const char* fixed = std::getenv("DRIVER_FIXED_ITERATIONS")) {  // a rate: exactly this many iterations
    solver.num_iterations = (size_t)std::atol(fixed);
    solver.absolute_error_tolerance = solver.relative_error_tolerance = 0.0;
  }
  if (mode == "jacobi" || mode == "jacobi-left") {
    solver.pre_op = std::make_unique<JacobiPreconditioner>();
    solver.pre_side = mode == "jacobi" ? PreconditionerSide::Right : PreconditionerSide::Left;
  }
  bool converged;
  ctx.sync();
  const auto t0 = std::chrono::steady_clock::now();
  if (native) {
    const HipStencilOperator op(matrix, -1.0, 0.0);  // A = -L
    converged = solver.solve(x, b, op);
    if (std::getenv("DRIVER_FIXED_ITERATIONS")) {  // (the first solve loaded the kernels: time a second one)
      DeviceVector x2(ctx, mesh.n_cells);
      ctx.sync();
      const auto t1 = std::chrono::steady_clock::now();
      solver.solve(x2, b, op);
      ctx.sync();
      std::printf("{\"timed_solve_seconds\": %.6f, \"timed_iterations\": %zu}\n",
                  std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count(), solver.iteration);
    }
  } else {
    const auto op = make_operator<DeviceVector>(
        [&](DeviceVector& y_vec, const DeviceVector& x_vec) { matrix.apply(-1.0, 0.0, x_vec, y_vec); });
    converged = solver.solve(x, b, *op);
  }
  ctx.sync();
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const std::vector<real_t> xh = x.to_host();
  const size_t c = ((size_t)(n / 2) * n + n / 2) * n + n / 2;
  std::printf("{\"n\": %d, \"converged\": %s, \"iterations\": %zu, \"absolute_error\": %.17g, "
              "\"relative_error\": %.17g, \"x_centre\": %.17g, \"x_norm2\": %.17g, \"x0\": %.17g, \"solve_seconds\": %.6f, "
              "\"lazy_cg_steps\": %lld, \"lazy_fused_dots\": %lld, \"lazy_apply_dots\": %lld}\n",
              n, converged ? "true" : "false", solver.iteration, solver.absolute_error, solver.relative_error,
              xh[c], norm_2(x), xh[0], seconds, ctx.counter("lazy_cg_steps"), ctx.counter("lazy_fused_dots"),
              ctx.counter("lazy_apply_dots"));
  if (logged.rfind("n_iter:", 0) != 0) return 4;  // Solver.hpp:144-145
  // error conventions: conj_mul of a plain operator throws std::runtime_error (Operator.hpp:116-118)
  try {
    make_operator<DeviceVector>([](DeviceVector&, const DeviceVector&) {})->conj_mul(x, b);
    return 3;
  } catch (const std::runtime_error&) {
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <n> <solver> <native|lambda|jacobi|jacobi-left> [restart]\n", argv[0]);
    return 2;
  }
  {  // knob names and defaults of the reference (Solver.hpp:66-76,158-159, SolverBiCgStab.hpp:379-381,
     // SolverIdrs.hpp:287-289, SolverRichardson.hpp:45)
    CgSolver<DeviceVector> cg;
    GmresSolver<DeviceVector> gm;
    BiCgStabLSolver<DeviceVector> bl;
    IdrsSolver<DeviceVector> id;
    RichardsonSolver<DeviceVector> ri;
    if (cg.num_iterations != 2000 || cg.absolute_error_tolerance != 1.0e-6 || cg.relative_error_tolerance != 1.0e-6 ||
        cg.pre_side != PreconditionerSide::Right || cg.pre_op != nullptr || gm.num_inner_iterations != 50 ||
        bl.num_inner_iterations != 2 || id.num_inner_iterations != 4 || ri.relaxation_factor != 1.0e-4)
      return 5;
  }
  const int n = std::atoi(argv[1]);
  const std::string kind = argv[2];
  const std::string native = argv[3];
  const size_t restart = argc > 4 ? (size_t)std::atoi(argv[4]) : 50;
  try {
    if (kind == "cg") return run<CgSolver>(n, native, restart);
    if (kind == "bicgstab") return run<BiCgStabSolver>(n, native, restart);
    if (kind == "gmres") return run<GmresSolver>(n, native, restart);
    if (kind == "fgmres") return run<FgmresSolver>(n, native, restart);
    if (kind == "jfnk") return run<JfnkSolver>(n, native, restart);
    if (kind == "bicgstabl") return run<BiCgStabLSolver>(n, native, restart);
    if (kind == "idrs") return run<IdrsSolver>(n, native, restart);
    if (kind == "cgs") return run<CgsSolver>(n, native, restart);
    if (kind == "tfqmr") return run<TfqmrSolver>(n, native, restart);
    if (kind == "tfqmr1") return run<Tfqmr1Solver>(n, native, restart);
    if (kind == "user-steepest-descent") return run<SteepestDescentSolver>(n, native, restart);
    if (kind == "user-cg") return run<StatementCgSolver>(n, native, restart);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 2;
}
