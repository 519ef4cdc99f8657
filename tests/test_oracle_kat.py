"""Pin the CPU oracle: the reference's own unit-test known answers (BLAS-1) and the reference
outputs recorded in BASELINE.md section 2 (solver loops + stencil)."""
import numpy as np
import pytest

from oracle import oracle
from stormruler_amd import mesh


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            flags = f.read()
        return " fma " in flags and " avx2 " in flags
    except OSError:
        return False


def test_real_matrix_reductions(golden):
    k = golden["unit_tests"]["real_matrix"]
    m = np.array(k["mat"])
    assert oracle.vsum(m) == k["sum"]
    assert oracle.norm1(m) == k["norm_1"]
    assert oracle.norm_inf(m) == k["norm_inf"]
    # CHECK_NEAR = doctest::Approx(expected).epsilon(eps)  (tests/unit/_UnitTests.hpp)
    assert abs(oracle.norm2(m) - k["norm_2"]) <= k["norm_2_eps"] * abs(k["norm_2"])


def test_dot_product(golden):
    k = golden["unit_tests"]["dot_product"]
    assert oracle.dot(k["mat1"], k["mat2"]) == k["dot"]
    assert oracle.dot(k["mat2"], k["mat1"]) == k["dot"]


def test_axpy_type_expression(golden):
    k = golden["unit_tests"]["expr_1"]
    out = oracle.expr1(k["mat1"], k["scale"], k["mat2"], k["mat3"])
    assert np.array_equal(out, np.array(k["result"]))


def test_product_and_quotient_expression(golden):
    """BitternMath.cpp:153-158 on the oracle's scalar loops (elementwise product in numpy: exact here)."""
    k = golden["unit_tests"]["expr_2"]
    t = -(np.array(k["mat1"]) * np.array(k["mat2"]))
    oracle.lib().oracle_mul_scalar(4, oracle._p(t), k["half"])
    q = np.array(k["mat3"])
    oracle.lib().oracle_div_scalar(4, oracle._p(q), k["hundredth"])
    oracle.lib().oracle_axpy(4, oracle._p(t), 1.0, oracle._p(q))
    assert np.array_equal(t, np.array(k["result"]))


def test_quotient_expressions(golden):
    """BitternMath.cpp:160-171 (expr-3, expr-4): scalar / matrix and matrix / matrix, exact on these values."""
    k = golden["unit_tests"]["expr_3"]
    m1, m2, m3 = (np.array(k[n]) for n in ("mat1", "mat2", "mat3"))
    t = 18.0 * m3  # 18 mat3 - 4 mat2 through the oracle's element loops
    oracle.lib().oracle_axmy(4, oracle._p(t), 4.0, oracle._p(m2))
    q = oracle.vdiv(24.0, None, m1)
    oracle.lib().oracle_axpy(4, oracle._p(q), 1.0, oracle._p(t))
    assert np.array_equal(q, np.array(k["result"]))
    k = golden["unit_tests"]["expr_4"]
    q = oracle.vdiv(9.0, m1, m3)
    oracle.lib().oracle_axmy(4, oracle._p(q), 1.0, oracle._p(m2))
    oracle.lib().oracle_mul_scalar(4, oracle._p(q), 2.0)
    assert np.array_equal(q, np.array(k["result"]))


def test_normalize_and_safe_divide(golden):
    k = golden["unit_tests"]["normalize"]
    m = np.array(k["mat"])
    assert oracle.norm2(m) == k["norm"]
    # normalize(0) == 0 pins safe_inverse / safe_divide (Crow/MathUtils.hpp:49-58)
    assert oracle.safe_divide(1.0, 0.0) == 0.0
    assert oracle.safe_divide(0.0, 0.0) == 0.0
    assert oracle.safe_divide(3.0, 2.0) == 1.5


def test_sym_ortho():
    cs, sn, rr = oracle.sym_ortho(3.0, 4.0)
    assert (cs, sn, rr) == (0.6, 0.8, 5.0)
    assert oracle.sym_ortho(0.0, 0.0) == (1.0, 0.0, 0.0)


def test_sequential_sum_order():
    # reduce() is a strict left-to-right sum (MatrixAlgorithms.hpp:191-205): 1e16 + 1 + 1 ... loses
    # every 1; a tree sum would not.
    a = np.array([1e16] + [1.0] * 8)
    assert oracle.dot(a, np.ones_like(a)) == 1e16


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_baseline_md_recorded_reference_outputs(golden, idx):
    case = golden["baseline_md_probe"]["cases"][idx]
    if not _cpu_has_fma():
        pytest.skip("recorded values were produced with FMA contraction; host has no FMA")
    n = case["n"]
    g = mesh.structured_box(n)
    g.validate()
    op = oracle.StencilOperator(g, alpha=-1.0, beta=0.0, variant="fma")
    r = oracle.solve(case["solver"], op, np.ones(g.n_cells), variant="fma",
                     num_inner_iterations=case.get("restart", 50))
    assert r.converged
    assert r.iterations == case["iterations"]
    assert r.num_applies == case["applies"]
    c = (n // 2 * n + n // 2) * n + n // 2
    # recorded to 7 / 13 significant digits
    assert abs(r.relative_error - case["rel"]) <= 5e-7 * case["rel"]
    assert abs(r.x[c] - case["x_centre"]) <= 5e-13 * abs(case["x_centre"])


@pytest.mark.parametrize("solver,restart", [("cg", 50), ("bicgstab", 50), ("gmres", 30)])
def test_strict_build_agrees_with_recorded_iteration_counts(golden, solver, restart):
    """The -ffp-contract=off build (the parity checker) gives the same iteration counts."""
    want = {c["solver"]: c for c in golden["baseline_md_probe"]["cases"] if c["n"] == 64 or c["solver"] == "cg"}
    n = 32 if solver == "cg" else 64
    if solver != "cg" and restart == 30:
        n = 64
    g = mesh.structured_box(n)
    op = oracle.StencilOperator(g, -1.0, 0.0)
    r = oracle.solve(solver, op, np.ones(g.n_cells), num_inner_iterations=restart)
    ref = [c for c in golden["baseline_md_probe"]["cases"] if c["solver"] == solver and c["n"] == n][0]
    assert r.iterations == ref["iterations"] and r.num_applies == ref["applies"]
    c = (n // 2 * n + n // 2) * n + n // 2
    assert abs(r.x[c] - ref["x_centre"]) <= 1e-6 * abs(ref["x_centre"])


def test_1d_poisson_closed_form():
    """-u'' = 1 on 64 points, Dirichlet via the 2/-1 stencil: x_31 = 528 exactly (SURVEY 8c KAT)."""
    import scipy.sparse as sp

    n = 64
    a = sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
    op = oracle.CsrOperator(a)
    for kind in ("cg", "gmres", "bicgstab"):
        r = oracle.solve(kind, op, np.ones(n), abs_tol=1e-10, rel_tol=1e-12)
        assert r.converged
        assert abs(r.x[31] - 528.0) < 1e-6, (kind, r.x[31])
        if kind in ("cg", "gmres"):
            assert r.iterations == 32


def test_face_loop_matches_assembled_matrix():
    g = mesh.structured_box(6, 5, 4)
    x = np.sin(0.37 * np.arange(g.n_cells))
    for alpha, beta in ((-1.0, 0.0), (-1e-2, 1.0)):
        y = oracle.StencilOperator(g, alpha, beta).apply(x)
        a = mesh.assemble_csr(g, alpha, beta)
        assert np.allclose(a @ x, y, rtol=0, atol=1e-12 * np.abs(y).max())
    # diagonal (6 + #walls)/h^2, off-diagonals -1/h^2 on the cube
    gc = mesh.structured_box(4)
    a = mesh.assemble_csr(gc, -1.0, 0.0).toarray()
    assert np.isclose(a[0, 0], 9 * 16) and np.isclose(a[0, 1], -16)
    i = (1 * 4 + 1) * 4 + 1
    assert np.isclose(a[i, i], 6 * 16)


def test_convergence_rule_edges():
    """Solver.hpp:124-140: early exit only on abs tol; tolerances <= 0 disable a test."""
    g = mesh.structured_box(8)
    op = oracle.StencilOperator(g, -1.0, 0.0)
    b = np.ones(g.n_cells)
    r = oracle.solve("cg", op, b, num_iterations=7, abs_tol=0.0, rel_tol=0.0)
    assert r.iterations == 7 and not r.converged
    r = oracle.solve("cg", op, b * 1e-9)  # initial error < abs tol -> 0 iterations, converged
    assert r.iterations == 0 and r.converged and r.num_applies == 1
    r = oracle.solve("cg", op, b, abs_tol=0.0, rel_tol=1e-3)
    assert r.converged and r.relative_error < 1e-3


def _tridiag(n=64):
    import scipy.sparse as sp

    return sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()


def test_preconditioned_gmres_branches():
    """SolverGmres.hpp pre_op != nullptr branches: an identity preconditioner on the left, and FGMRES with
    one, repeat the unpreconditioned arithmetic bit for bit; on the right (non-flexible) x is assembled
    through q_0 (:242-247) so only the values agree.  All reach the 1-D KAT x_31 = 528."""
    a = _tridiag()
    op, b = oracle.CsrOperator(a), np.ones(64)
    ident = oracle.DiagOperator(np.ones(64))
    plain = oracle.solve("gmres", op, b, num_inner_iterations=50)
    assert plain.iterations == 32 and abs(plain.x[31] - 528.0) < 1e-6
    for side, flexible, exact in (("left", False, True), ("right", True, True), ("left", True, True),
                                  ("right", False, False)):
        r, n_pre = oracle.solve_gmres_pre(op, ident, b, side=side, flexible=flexible, num_inner_iterations=50)
        assert r.iterations == plain.iterations and np.array_equal(r.history, plain.history)
        assert np.array_equal(r.x, plain.x) if exact else np.allclose(r.x, plain.x, rtol=1e-12)
        # applications: left = 1 per start + 1 per iteration; flexible = 1 per iteration;
        # right = 1 per iteration + 1 per finalize
        starts = 2  # outer_init + inner_init of the first cycle (SolverGmres.hpp:66-67 duplication)
        want = {("left", False): starts + r.iterations, ("right", False): r.iterations + 1}.get((side, flexible),
                                                                                               r.iterations)
        assert n_pre == want
    # symmetric side: neither branch is taken (SolverGmres.hpp:121-128) -> plain GMRES
    r, n_pre = oracle.solve_gmres_pre(op, ident, b, side="symmetric", num_inner_iterations=50)
    assert n_pre == 0 and np.array_equal(r.x, plain.x)
    # a real preconditioner on a badly scaled system cuts the iteration count
    d = np.linspace(1.0, 1e3, 64)
    import scipy.sparse as sp

    op2 = oracle.CsrOperator((sp.diags(d) @ a).tocsr())
    bad = oracle.solve("gmres", op2, d, num_inner_iterations=50, num_iterations=500)
    good, _ = oracle.solve_gmres_pre(op2, oracle.DiagOperator(1.0 / d), d, side="left", num_inner_iterations=50,
                                     num_iterations=500)
    assert good.converged and good.iterations == 32  # D^-1 (D A) = A again
    assert (not bad.converged) or bad.iterations > good.iterations
    assert abs(good.x[31] - 528.0) < 1e-3


def test_jfnk_linear_and_nonlinear():
    """SolverNewton.hpp:101-173.  On a linear operator the finite-difference Jacobian is exact up to rounding,
    so one Newton step (one inner BiCGStab solve to 1e-8) lands on the KAT; on a cubic perturbation the
    residual history contracts quadratically near the root."""
    a = _tridiag()
    r, inner = oracle.solve_jfnk(oracle.CsrOperator(a), np.ones(64))
    assert r.converged and r.iterations <= 2 and abs(r.x[31] - 528.0) < 1e-4
    assert inner >= 32 and r.num_applies >= inner * 2
    shifted = (a + 4.0 * __import__("scipy.sparse").sparse.identity(64)).tocsr()
    f = lambda v: shifted @ v + 0.2 * v ** 3  # noqa: E731
    r, inner = oracle.solve_jfnk(oracle.CallbackOperator(64, f), np.ones(64))
    assert r.converged and 2 <= r.iterations <= 8
    assert np.abs(f(r.x) - 1.0).max() < 1e-6
    h = r.history
    assert h[-1] < 1e-3 * h[-2] or h[-1] < 1e-9  # fast final contraction


@pytest.mark.parametrize("kind", ["cg", "bicgstab", "cgs", "tfqmr", "tfqmr1", "bicgstabl", "idrs", "gmres", "richardson"])
def test_preconditioned_branches_of_every_solver(kind):
    """pre_op / pre_side (Solver.hpp:74-75) in every restated driver: an identity preconditioner on the LEFT
    repeats the unpreconditioned arithmetic bit for bit (the left branches only insert `r <- P r`), and a real
    diagonal preconditioner on a badly row-scaled system converges to the 1-D KAT from either side."""
    import scipy.sparse as sp

    n = 64
    a = _tridiag(n)
    b = np.ones(n)
    kw = {"num_inner_iterations": {"bicgstabl": 2, "idrs": 4}.get(kind, 50)}
    if kind == "richardson":
        kw.update(relaxation_factor=0.4, num_iterations=300, abs_tol=0.0, rel_tol=0.0)
    op = oracle.CsrOperator(a)
    oracle.lib().oracle_rng_reset()
    plain = oracle.solve(kind, op, b, **kw)
    oracle.lib().oracle_rng_reset()
    ident = oracle.solve(kind, op, b, pre=oracle.DiagOperator(np.ones(n)), side="left", **kw)
    assert ident.iterations == plain.iterations and np.array_equal(ident.x, plain.x)
    assert oracle.last_pre_applies() > 0 or kind == "gmres"
    if kind == "richardson":
        return
    d = np.linspace(1.0, 200.0, n)
    scaled = (sp.diags(d) @ a).tocsr() if kind != "cg" else (sp.diags(np.sqrt(d)) @ a @ sp.diags(np.sqrt(d))).tocsr()
    rhs = d.copy() if kind != "cg" else np.sqrt(d) * b
    sop = oracle.CsrOperator(scaled)
    pre = oracle.DiagOperator(1.0 / scaled.diagonal())
    for side in ("left", "right"):
        oracle.lib().oracle_rng_reset()
        r = oracle.solve(kind, sop, rhs, pre=pre, side=side, abs_tol=1e-9, rel_tol=1e-11, **kw)
        assert r.converged, (kind, side)
        x = r.x if kind != "cg" else np.sqrt(d) * r.x   # undo the symmetric scaling
        assert abs(x[31] - 528.0) < 1e-5 * 528.0, (kind, side, x[31])


def test_openmp_port_agrees_with_the_sequential_oracle():
    """oracle/storm_oracle_omp.c (bench.py's optional parallel CPU line) is the same mathematics: the residual
    after K CG steps on the Dirichlet box equals the sequential port's to rounding, for any thread count."""
    from stormruler_amd import mesh

    n, k = 20, 25
    g = mesh.structured_box(n)
    r = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells), num_iterations=k, abs_tol=0.0,
                     rel_tol=0.0)
    for threads in (1, 3):
        res, sec, used = oracle.omp_cg_box(n, k, threads)
        assert used == threads and sec > 0.0
        assert abs(res - r.absolute_error) <= 1e-10 * r.absolute_error
    a = mesh.assemble_csr(g, -1.0, 0.0)
    x, res, _ = oracle.omp_cg(a, np.ones(g.n_cells), k, 2)
    assert abs(res - r.absolute_error) <= 1e-10 * r.absolute_error
    assert np.abs(x - r.x).max() <= 1e-10 * np.abs(r.x).max()
