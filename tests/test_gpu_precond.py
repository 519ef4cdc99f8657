"""SURVEY.md 8f rank 3, the rest of it: the pre_op / pre_side hook with a real (device-side Jacobi)
preconditioner, preconditioned GMRES, flexible GMRES with a varying preconditioner, and JFNK --
each against the oracle's restatement of the same reference branches (SolverGmres.hpp, SolverNewton.hpp)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NU, VEL = 1e-2, (1.0, 0.5, 0.25)


@pytest.fixture(scope="module")
def env():
    from oracle import oracle
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    yield api, mesh, oracle, ctx
    ctx.close()


def _graded_box(mesh, n):
    """A box whose cell volumes vary (so the diagonal does, and Jacobi actually changes the iteration)."""
    g = mesh.structured_box(n)
    rng = np.random.default_rng(7)
    g.volume = g.volume * (0.25 + 1.5 * rng.random(g.n_total))
    return g


@pytest.mark.parametrize("build", ["faces", "weights", "csr", "faces_tail"])
def test_diagonal_matches_assembled_matrix(env, build):
    api, mesh, oracle, ctx = env
    g = _graded_box(mesh, 13)
    alpha, beta = -0.7, 0.3
    if build == "weights":
        wi, wo, de = mesh.convection_diffusion_weights(g, NU, VEL)
        mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
        a = None  # no assembled cross-check for the upwind weights: probe (A e_i)_i instead
    elif build == "csr":
        a = mesh.assemble_csr(g, 1.0, 0.0)
        mat = api.StencilMatrix.from_csr(ctx, a)
    else:
        if build == "faces_tail":
            ctx.set_option("ell_cap", 3)  # rows with > 3 neighbours overflow into the CSR tail
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        ctx.set_option("ell_cap", 0)
        if build == "faces_tail":
            assert mat.stats()["tail_rows"] > 0
        a = mesh.assemble_csr(g, 1.0, 0.0)
    if a is None:  # diagonal by probing: (A e_i)_i for a handful of rows
        x, y = api.DeviceVector(ctx, g.n_cells), api.DeviceVector(ctx, g.n_cells)
        rows = [0, 5, g.n_cells // 2, g.n_cells - 1]
        want = {}
        for r in rows:
            e = np.zeros(g.n_cells)
            e[r] = 1.0
            x.upload(e)
            mat.apply(alpha, beta, x, y)
            want[r] = y.to_numpy()[r]
    d = api.DeviceVector(ctx, g.n_cells)
    mat.diagonal(alpha, beta, d)
    got = d.to_numpy()
    if a is not None:
        ref = beta + alpha * a.diagonal()
        assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    else:
        for r, v in want.items():
            assert abs(got[r] - v) <= 1e-13 * abs(v)
    mat.diagonal(alpha, beta, d, invert=True)
    assert np.abs(d.to_numpy() * got - 1.0).max() <= 4e-16
    mat.close()


def test_vmul_and_safe_inverse(env):
    api, mesh, oracle, ctx = env
    n = 4099  # odd: exercises the scalar tail of the 16-byte kernel
    a, b = np.sin(0.3 * np.arange(n)), np.cos(0.11 * np.arange(n))
    av, bv, yv = (api.DeviceVector.from_numpy(ctx, a), api.DeviceVector.from_numpy(ctx, b), api.DeviceVector(ctx, n))
    api.vmul(yv, av, bv)
    assert np.array_equal(yv.to_numpy(), a * b)
    api.vmul(av, av, bv)  # aliasing allowed
    assert np.array_equal(av.to_numpy(), a * b)
    # a zero diagonal entry inverts to zero (safe_inverse): identity-free operator beta = alpha = 0
    g = mesh.structured_box(5)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    d = api.DeviceVector(ctx, g.n_cells)
    mat.diagonal(0.0, 0.0, d, invert=True)
    assert np.array_equal(d.to_numpy(), np.zeros(g.n_cells))
    mat.close()


def _problem(env, n=20):
    api, mesh, oracle, ctx = env
    g = _graded_box(mesh, n)
    wi, wo, de = mesh.convection_diffusion_weights(g, NU, VEL)
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    op = api.HipStencilOperator(mat, 1.0, 0.0)
    ref_op = oracle.StencilOperator(g, -NU, 0.0, conv=1.0, vel=VEL)
    d = api.DeviceVector(ctx, g.n_cells)
    mat.diagonal(1.0, 0.0, d, invert=True)
    return g, mat, op, ref_op, d.to_numpy()


@pytest.mark.parametrize("side", ["left", "right"])
@pytest.mark.parametrize("flexible", [False, True])
def test_jacobi_preconditioned_gmres_matches_oracle(env, side, flexible):
    api, mesh, oracle, ctx = env
    g, mat, op, ref_op, dinv = _problem(env)
    b_host = np.ones(g.n_cells)
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
    s = api.FgmresSolver() if flexible else api.GmresSolver()
    s.num_inner_iterations = 20
    s.pre_op = api.JacobiPreconditioner()
    s.pre_side = api.PreconditionerSide.Left if side == "left" else api.PreconditionerSide.Right
    s.record_history = True
    assert s.solve(x, b, op)
    ref, n_pre = oracle.solve_gmres_pre(ref_op, oracle.DiagOperator(dinv), b_host, side=side, flexible=flexible,
                                        num_inner_iterations=20)
    plain = oracle.solve("gmres", ref_op, b_host, num_inner_iterations=20)
    assert ref.converged and ref.iterations != plain.iterations  # the preconditioner changes the iteration
    assert abs(s.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations))
    assert np.linalg.norm(x.to_numpy() - ref.x) <= 1e-6 * np.linalg.norm(ref.x)
    m = min(len(s.history), len(ref.history), 10)
    assert np.allclose(s.history[:m], ref.history[:m], rtol=1e-8)
    mat.close()


def test_flexible_gmres_with_a_varying_preconditioner(env):
    """What FGMRES exists for (SolverGmres.hpp:285-308): P changes at every application."""
    api, mesh, oracle, ctx = env
    g, mat, op, ref_op, dinv = _problem(env)
    b_host = np.ones(g.n_cells)
    scale = lambda k: 1.0 + 0.5 * ((k * 7) % 5) / 5.0  # noqa: E731

    class Varying(api.Preconditioner):
        def __init__(self):
            self.calls = 0

        def build(self, x_vec, b_vec, any_op):
            self.d = api.DeviceVector.from_numpy(ctx, dinv)

        def mul(self, y_vec, x_vec):
            api.vmul(y_vec, self.d, x_vec)
            y_vec *= scale(self.calls)
            self.calls += 1

    calls = [0]

    def ref_pre(v):
        out = (dinv * v) * scale(calls[0])
        calls[0] += 1
        return out

    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
    s = api.FgmresSolver()
    s.num_inner_iterations = 15
    s.pre_op = Varying()
    assert s.solve(x, b, op)
    ref, n_pre = oracle.solve_gmres_pre(ref_op, oracle.CallbackOperator(g.n_cells, ref_pre), b_host, flexible=True,
                                        num_inner_iterations=15)
    # the device loop enqueues a few iterations ahead of the convergence verdict: the callback is ENTERED more
    # often than the preconditioner is applied; the logical count is the reference's
    assert ref.converged and n_pre == s.num_pre_applies == ref.iterations and s.pre_op.calls >= n_pre
    assert abs(s.iteration - ref.iterations) <= 2
    assert np.linalg.norm(x.to_numpy() - ref.x) <= 1e-6 * np.linalg.norm(ref.x)
    # the true residual of the returned x honours the reported (right-preconditioned => true) norm
    r = api.DeviceVector(ctx, g.n_cells)
    op.Residual(r, b, x)
    assert abs(api.norm_2(r) - s.absolute_error) <= 1e-6 * np.linalg.norm(b_host)
    mat.close()


@pytest.mark.parametrize("cls", ["CgSolver", "BiCgStabSolver"])
def test_jacobi_hook_on_the_poisson_operator(env, cls):
    api, mesh, oracle, ctx = env
    # V^-1 K with varying volumes is not symmetric: CG gets the uniform box (diagonal 6, 7, 8, 9 / h^2)
    g = mesh.structured_box(16) if cls == "CgSolver" else _graded_box(mesh, 16)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells))
    x0, x1 = api.DeviceVector(ctx, g.n_cells), api.DeviceVector(ctx, g.n_cells)
    plain, pre = getattr(api, cls)(), getattr(api, cls)()
    assert plain.solve(x0, b, op)
    pre.pre_op = api.JacobiPreconditioner()
    assert pre.solve(x1, b, op)
    assert np.linalg.norm(x1.to_numpy() - x0.to_numpy()) <= 1e-5 * np.linalg.norm(x0.to_numpy())
    mat.close()


def test_jfnk_on_a_nonlinear_operator(env):
    """A(x) = x - kappa L x + c x^3 (a cubic reaction term like the playground's dF/dc, Playground.cpp:140-148)."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(12)
    kappa, c3 = 1e-2, 0.5
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    sq = api.DeviceVector(ctx, g.n_cells)

    def nonlinear(y, x):
        mat.apply(-kappa, 1.0, x, y)
        api.vmul(sq, x, x)
        api.vmul_add(y, c3, sq, x)

    lin = oracle.StencilOperator(g, -kappa, 1.0)
    ref_op = oracle.CallbackOperator(g.n_cells, lambda v: lin.apply(v) + c3 * ((v * v) * v))
    b_host = 1.0 + 0.5 * np.sin(5.0 * g.center[: g.n_cells, 0])
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
    s = api.JfnkSolver()
    s.record_history = True
    assert s.solve(x, b, api.make_operator(nonlinear))
    ref, inner = oracle.solve_jfnk(ref_op, b_host)
    assert ref.converged and 1 < ref.iterations < 20
    assert s.iteration == ref.iterations
    assert abs(s.inner_iterations - inner) <= max(2, int(0.1 * inner))
    assert np.linalg.norm(x.to_numpy() - ref.x) <= 1e-7 * np.linalg.norm(ref.x)
    res = lin.apply(x.to_numpy()) + c3 * x.to_numpy() ** 3 - b_host
    assert np.linalg.norm(res) < 1.01 * max(1e-6, 1e-6 * ref.initial_error)
    with pytest.raises(NotImplementedError):  # SolverNewton.hpp:61: declared, unimplemented
        api.NewtonSolver().solve(x, b, api.make_operator(nonlinear))
    mat.close()


def _run(*args):
    exe = os.path.join(ROOT, "tests", "cpp", "poisson_driver")
    out = subprocess.run([exe, *map(str, args)], check=True, capture_output=True, text=True, timeout=600).stdout
    return json.loads(out.strip().splitlines()[-1])


@pytest.mark.parametrize("kind,mode,side,flexible", [("gmres", "jacobi", "right", False),
                                                     ("gmres", "jacobi-left", "left", False),
                                                     ("fgmres", "jacobi", "right", True)])
def test_cpp_adapter_preconditioned_gmres(kind, mode, side, flexible):
    from oracle import oracle
    from stormruler_amd import mesh

    n, restart = 20, 25
    got = _run(n, kind, mode, restart)
    g = mesh.structured_box(n)
    a = mesh.assemble_csr(g, -1.0, 0.0)
    ref, _ = oracle.solve_gmres_pre(oracle.StencilOperator(g, -1.0, 0.0), oracle.DiagOperator(1.0 / a.diagonal()),
                                    np.ones(g.n_cells), side=side, flexible=flexible, num_inner_iterations=restart)
    assert got["converged"] and ref.converged
    assert abs(got["iterations"] - ref.iterations) <= max(2, int(0.05 * ref.iterations))
    assert abs(got["x_norm2"] - np.linalg.norm(ref.x)) <= 1e-6 * np.linalg.norm(ref.x)


def test_cpp_adapter_jfnk_and_fgmres_native():
    from oracle import oracle
    from stormruler_amd import mesh

    n = 16
    g = mesh.structured_box(n)
    ref_op = oracle.StencilOperator(g, -1.0, 0.0)
    got = _run(n, "jfnk", "native")
    ref, _ = oracle.solve_jfnk(ref_op, np.ones(g.n_cells))
    assert got["converged"] and got["iterations"] == ref.iterations
    assert abs(got["x_norm2"] - np.linalg.norm(ref.x)) <= 1e-7 * np.linalg.norm(ref.x)
    got = _run(n, "fgmres", "native", 30)  # no preconditioner: FGMRES == GMRES, native device path
    ref = oracle.solve("gmres", ref_op, np.ones(g.n_cells), num_inner_iterations=30)
    assert got["converged"] and abs(got["iterations"] - ref.iterations) <= 2
    assert abs(got["x_norm2"] - np.linalg.norm(ref.x)) <= 1e-6 * np.linalg.norm(ref.x)


ALL_KINDS = [("cg", "CgSolver"), ("bicgstab", "BiCgStabSolver"), ("cgs", "CgsSolver"), ("tfqmr", "TfqmrSolver"),
             ("tfqmr1", "Tfqmr1Solver"), ("bicgstabl", "BiCgStabLSolver"), ("idrs", "IdrsSolver"),
             ("richardson", "RichardsonSolver")]


@pytest.mark.parametrize("side", ["left", "right"])
@pytest.mark.parametrize("kind,cls", ALL_KINDS)
def test_every_solver_with_jacobi_matches_oracle(env, kind, cls, side):
    """The left / right preconditioned branches of every driver (the reference's `pre_op` / `pre_side` members,
    Solver.hpp:74-75) against the oracle's restatement of the same branches, with the device-side Jacobi."""
    api, mesh, oracle, ctx = env
    spd = kind in ("cg", "richardson")
    if spd:  # uniform box: symmetric; its diagonal still varies (6, 7, 8, 9 / h^2)
        g = mesh.structured_box(14)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        op, ref_op = api.HipStencilOperator(mat, -1.0, 0.0), oracle.StencilOperator(g, -1.0, 0.0)
        alpha_beta = (-1.0, 0.0)
    else:
        g, mat, op, ref_op, _ = _problem(env, 14)
        alpha_beta = (1.0, 0.0)
    d = api.DeviceVector(ctx, g.n_cells)
    mat.diagonal(alpha_beta[0], alpha_beta[1], d, invert=True)
    dinv = d.to_numpy()
    b_host = np.ones(g.n_cells)
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
    s = getattr(api, cls)()
    kw = {}
    if kind == "richardson":  # omega = 1 with Jacobi = damped Jacobi iteration; converges slowly but surely
        s.relaxation_factor, kw["relaxation_factor"] = 0.9, 0.9
        s.num_iterations, kw["num_iterations"] = 4000, 4000
        s.relative_error_tolerance = s.absolute_error_tolerance = 1e-3
        kw["abs_tol"] = kw["rel_tol"] = 1e-3
    if kind in ("bicgstabl", "idrs"):
        kw["num_inner_iterations"] = s.num_inner_iterations
    s.pre_op = api.JacobiPreconditioner()
    s.pre_side = api.PreconditionerSide.Left if side == "left" else api.PreconditionerSide.Right
    api.rng_reset()
    oracle.lib().oracle_rng_reset()
    s.record_history = True
    assert s.solve(x, b, op)
    ref = oracle.solve(kind, ref_op, b_host, pre=oracle.DiagOperator(dinv), side=side, **kw)
    assert ref.converged
    tol_it = max(2, int(0.1 * ref.iterations))  # the short recurrences amplify rounding differences
    assert abs(s.iteration - ref.iterations) <= tol_it, (s.iteration, ref.iterations)
    assert np.linalg.norm(x.to_numpy() - ref.x) <= (2e-3 if kind == "richardson" else 2e-5) * np.linalg.norm(ref.x)
    m = min(len(s.history), len(ref.history), 6)
    assert np.allclose(s.history[:m], ref.history[:m], rtol=1e-6)
    mat.close()


@pytest.mark.parametrize("kind", ["cg", "bicgstab", "cgs", "tfqmr", "tfqmr1", "bicgstabl", "idrs"])
@pytest.mark.parametrize("mode,side", [("jacobi", "right"), ("jacobi-left", "left")])
def test_cpp_adapter_every_solver_with_jacobi(kind, mode, side):
    """The same branches in the C++ restatement of the class templates (include/storm_hip/Storm.hpp)."""
    from oracle import oracle
    from stormruler_amd import mesh

    n = 16
    restart = {"bicgstabl": 2, "idrs": 4}.get(kind, 50)
    got = _run(n, kind, mode, restart)
    g = mesh.structured_box(n)
    a = mesh.assemble_csr(g, -1.0, 0.0)
    oracle.lib().oracle_rng_reset()
    ref = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells),
                       pre=oracle.DiagOperator(1.0 / a.diagonal()), side=side, num_inner_iterations=restart)
    assert got["converged"] and ref.converged
    assert abs(got["iterations"] - ref.iterations) <= max(2, int(0.1 * ref.iterations)), (got["iterations"], ref.iterations)
    assert abs(got["x_norm2"] - np.linalg.norm(ref.x)) <= 1e-5 * np.linalg.norm(ref.x)


@pytest.mark.parametrize("seed", range(5))
def test_random_matrices_every_solver_matches_oracle(env, seed):
    """Randomised parity: diagonally dominant sparse matrices (symmetric for CG, non-symmetric otherwise) with
    random patterns and values (fp64 records), every driver -- native device loops and statement-level ones --
    against the oracle on the same CSR rows."""
    import scipy.sparse as sp

    api, mesh, oracle, ctx = env
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(200, 1500))
    nnz_row = int(rng.integers(2, 9))
    rows = np.repeat(np.arange(n), nnz_row)
    cols = (rows + rng.integers(1, n, size=rows.size)) % n
    off = sp.coo_matrix((rng.uniform(-1.0, 1.0, rows.size), (rows, cols)), shape=(n, n)).tocsr()
    off.sum_duplicates()
    nonsym = off + sp.diags(np.abs(off).sum(axis=1).A1 * 1.5 + 1.0)
    sym_off = (off + off.T) * 0.5
    sym = sym_off + sp.diags(np.abs(sym_off).sum(axis=1).A1 * 1.5 + 1.0)
    b_host = rng.standard_normal(n)
    for kind, cls, a in (("cg", "CgSolver", sym), ("bicgstab", "BiCgStabSolver", nonsym), ("gmres", "GmresSolver", nonsym),
                         ("cgs", "CgsSolver", nonsym), ("tfqmr", "TfqmrSolver", nonsym), ("tfqmr1", "Tfqmr1Solver", nonsym),
                         ("bicgstabl", "BiCgStabLSolver", nonsym), ("idrs", "IdrsSolver", nonsym)):
        a = a.tocsr()
        mat = api.StencilMatrix.from_csr(ctx, a)
        op = api.HipStencilOperator(mat, 1.0, 0.0)
        b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, n)
        s = getattr(api, cls)()
        api.rng_reset()
        oracle.lib().oracle_rng_reset()
        assert s.solve(x, b, op), kind
        ref = oracle.solve(kind, oracle.CsrOperator(a), b_host, num_inner_iterations=s.num_inner_iterations
                           if hasattr(s, "num_inner_iterations") else 50)
        assert ref.converged
        assert abs(s.iteration - ref.iterations) <= max(2, int(0.1 * ref.iterations)), (kind, s.iteration, ref.iterations)
        assert np.linalg.norm(x.to_numpy() - ref.x) <= 1e-7 * np.linalg.norm(ref.x), kind
        assert np.abs(a @ x.to_numpy() - b_host).max() <= 1e-4 * np.abs(b_host).max()
        mat.close()
