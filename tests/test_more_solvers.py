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
