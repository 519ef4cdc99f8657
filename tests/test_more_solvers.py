"""the other Krylov drivers on the same kernels -- CGS, TFQMR, TFQMR1, Richardson.
Oracle restatements pinned by the 1-D known answer (x_31 = 528 in 32 iterations for CGS / TFQMR1, as
SURVEY 8c records from the reference's own templates), then HIP-vs-oracle parity on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle
from stormruler_amd import mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_1d_poisson_known_answer():
    n = 64
    a = sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsr()
    op = oracle.CsrOperator(a)
    for kind, iters in (("cgs", 32), ("tfqmr1", 32), ("tfqmr", 32)):
        r = oracle.solve(kind, op, np.ones(n), abs_tol=1e-10, rel_tol=1e-12)
        assert r.converged and r.iterations == iters
        assert abs(r.x[31] - 528.0) < 1e-9


def test_oracle_solvers_on_the_3d_stencil():
    g = mesh.structured_box(12)
    op = oracle.StencilOperator(g, -1.0, 0.0)
    b = np.ones(g.n_cells)
    ref = oracle.solve("cg", op, b, rel_tol=1e-10, abs_tol=0.0)
    for kind in ("cgs", "tfqmr", "tfqmr1"):
        r = oracle.solve(kind, op, b)
        assert r.converged and np.linalg.norm(r.x - ref.x) <= 1e-5 * np.linalg.norm(ref.x)
        assert r.num_applies == 1 + 2 * r.iterations
    # Richardson converges for omega < 2 / lambda_max ~ 2 h^2 / 12
    r = oracle.solve("richardson", op, b, relaxation_factor=1.0 / (12 * 144), num_iterations=5000)
    assert r.converged and r.num_applies == 1 + r.iterations
    assert np.linalg.norm(r.x - ref.x) <= 1e-4 * np.linalg.norm(ref.x)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cgs", "tfqmr", "tfqmr1", "richardson"])
def test_hip_statement_path_matches_oracle(kind):
    from stormruler_amd import api

    g = mesh.structured_box(20, 16, 12)
    ctx = api.Context(0)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = np.cos(0.01 * np.arange(g.n_cells)) + 0.5
    cls = {"cgs": api.CgsSolver, "tfqmr": api.TfqmrSolver, "tfqmr1": api.Tfqmr1Solver,
           "richardson": api.RichardsonSolver}[kind]
    s = cls()
    kw = {}
    if kind == "richardson":
        s.relaxation_factor = kw["relaxation_factor"] = 1.0 / (12 * 400)
        s.num_iterations = kw["num_iterations"] = 300
        s.relative_error_tolerance = kw["rel_tol"] = 1e-2
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
    s.record_history = True
    ok = s.solve(x, b, op)
    ref = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), b_host, **kw)
    assert ok == ref.converged
    assert abs(s.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations))
    assert np.linalg.norm(x.to_numpy() - ref.x) <= 5e-6 * np.linalg.norm(ref.x)
    m = min(len(s.history), len(ref.history), 8)
    assert np.allclose(s.history[:m], ref.history[:m], rtol=1e-8)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cgs", "tfqmr", "tfqmr1"])
def test_cpp_adapter_solvers(kind):
    driver = os.path.join(ROOT, "tests", "cpp", "poisson_driver")
    if not os.path.exists(driver):
        import __graft_entry__ as ge

        ge.build()
    out = subprocess.run([driver, "20", kind, "lambda"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])
    g = mesh.structured_box(20)
    ref = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells))
    assert got["converged"] and abs(got["iterations"] - ref.iterations) <= max(2, int(0.05 * ref.iterations))
    assert abs(got["x_norm2"] - np.linalg.norm(ref.x)) <= 5e-6 * np.linalg.norm(ref.x)


# ---- BiCGStab(l), IDR(s), fill_randomly -----------------------------------------------------------

def test_oracle_mt19937_64_matches_the_standard():
    """The C++ standard fixes the 10000th output of a default-constructed mt19937_64."""
    L = oracle.lib()
    oracle.rng_reset()
    for _ in range(9999):
        L.oracle_rng_next()
    assert L.oracle_rng_next() == 9981545732273789042
    oracle.rng_reset()
    v = oracle.fill_randomly(1000)
    assert np.all((v >= 0.0) & (v < 1.0)) and abs(v.mean() - 0.5) < 0.05


def test_oracle_bicgstabl_and_idrs_converge():
    g = mesh.structured_box(12)
    op = oracle.StencilOperator(g, -1.0, 0.0)
    b = np.ones(g.n_cells)
    ref = oracle.solve("cg", op, b, rel_tol=1e-10, abs_tol=0.0)
    for kind, m in (("bicgstabl", 2), ("bicgstabl", 4), ("idrs", 4), ("idrs", 1), ("idrs", 8)):
        oracle.rng_reset()
        r = oracle.solve(kind, op, b, num_inner_iterations=m)
        assert r.converged and np.linalg.norm(r.x - ref.x) <= 1e-5 * np.linalg.norm(ref.x), (kind, m)
    # IDR(s) consumes the function-static engine: a second solve in the same "process" sees other numbers
    oracle.rng_reset()
    r1 = oracle.solve("idrs", op, b, num_inner_iterations=4)
    r2 = oracle.solve("idrs", op, b, num_inner_iterations=4)
    oracle.rng_reset()
    r3 = oracle.solve("idrs", op, b, num_inner_iterations=4)
    assert np.array_equal(r1.x, r3.x) and not np.array_equal(r1.x, r2.x)


@pytest.mark.gpu
def test_fill_randomly_is_the_reference_sequence():
    from stormruler_amd import api

    ctx = api.Context(0)
    api.rng_reset()
    oracle.rng_reset()
    a, b = api.DeviceVector(ctx, 1000), api.DeviceVector(ctx, 37)
    api.fill_randomly(a)
    api.fill_randomly(b)  # the engine's state persists across calls (function-static in the reference)
    want = oracle.fill_randomly(1037)
    assert np.array_equal(a.to_numpy(), want[:1000]) and np.array_equal(b.to_numpy(), want[1000:])
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,m", [("bicgstabl", 2), ("bicgstabl", 3), ("idrs", 4), ("idrs", 2)])
def test_hip_bicgstabl_idrs_match_oracle(kind, m):
    from stormruler_amd import api

    g = mesh.structured_box(18, 14, 11)
    ctx = api.Context(0)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = np.cos(0.01 * np.arange(g.n_cells)) + 0.5
    s = (api.BiCgStabLSolver if kind == "bicgstabl" else api.IdrsSolver)()
    s.num_inner_iterations = m
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
    api.rng_reset()
    oracle.rng_reset()
    s.record_history = True
    ok = s.solve(x, b, op)
    ref = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), b_host, num_inner_iterations=m)
    assert ok and ref.converged
    assert abs(s.iteration - ref.iterations) <= max(2, int(0.1 * ref.iterations))
    assert np.linalg.norm(x.to_numpy() - ref.x) <= 1e-5 * np.linalg.norm(ref.x)
    k = min(len(s.history), len(ref.history), 6)
    assert np.allclose(s.history[:k], ref.history[:k], rtol=1e-7)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,m", [("bicgstabl", 2), ("idrs", 4)])
def test_cpp_adapter_bicgstabl_idrs(kind, m):
    driver = os.path.join(ROOT, "tests", "cpp", "poisson_driver")
    out = subprocess.run([driver, "16", kind, "lambda", str(m)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])
    g = mesh.structured_box(16)
    oracle.rng_reset()
    ref = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells), num_inner_iterations=m)
    assert got["converged"] and abs(got["iterations"] - ref.iterations) <= max(2, int(0.1 * ref.iterations))
    assert abs(got["x_norm2"] - np.linalg.norm(ref.x)) <= 1e-5 * np.linalg.norm(ref.x)


@pytest.mark.gpu
def test_preconditioner_hook_sides():
    """pre_op / pre_side (Solver.hpp:74-75): the identity preconditioner must not change the iterates;
    a Jacobi-like diagonal scaling (one elementwise product per application) must still converge."""
    from stormruler_amd import api

    g = mesh.structured_box(12)
    ctx = api.Context(0)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    lam = api.make_operator(lambda y, x: mat.apply(-1.0, 0.0, x, y))
    b = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells))
    base = {}
    for cls in (api.CgSolver, api.BiCgStabSolver):
        x = api.DeviceVector(ctx, g.n_cells)
        s = cls()
        assert s.solve(x, b, lam)
        base[cls] = (s.iteration, x.to_numpy())
    for cls in (api.CgSolver, api.BiCgStabSolver):
        for side in (api.PreconditionerSide.Right, api.PreconditionerSide.Left):
            x = api.DeviceVector(ctx, g.n_cells)
            s = cls()
            s.pre_op, s.pre_side = api.IdentityPreconditioner(), side
            assert s.solve(x, b, lam)
            assert s.iteration == base[cls][0]
            assert np.linalg.norm(x.to_numpy() - base[cls][1]) <= 1e-12 * np.linalg.norm(base[cls][1])
    # diagonal preconditioner M^-1 = 1/diag(A)
    import scipy.sparse as sp  # noqa: F401
    diag = mesh.assemble_csr(g, -1.0, 0.0).diagonal()
    dinv = api.DeviceVector.from_numpy(ctx, 1.0 / diag)

    class Jacobi(api.Preconditioner):
        def mul(self, y, x):
            api.fill_with(y, 0.0)
            api.vmul_add(y, 1.0, dinv, x)

    x = api.DeviceVector(ctx, g.n_cells)
    s = api.CgSolver()
    s.pre_op = Jacobi()
    assert s.solve(x, b, lam)
    assert np.linalg.norm(x.to_numpy() - base[api.CgSolver][1]) <= 1e-5 * np.linalg.norm(base[api.CgSolver][1])
    ctx.close()
