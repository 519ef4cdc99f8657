"""BASELINE config 4: GMRES(30) on the first-order upwind convection-diffusion operator
A = -nu L + C(v), nu = 1e-2, v = (1, 0.5, 0.25), Dirichlet 0 -- the non-symmetric face-weight path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NU, VEL = 1e-2, (1.0, 0.5, 0.25)


@pytest.fixture(scope="module")
def env():
    from oracle import oracle
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    yield api, mesh, oracle, ctx
    ctx.close()


def _matrix(api, mesh, ctx, g):
    wi, wo, de = mesh.convection_diffusion_weights(g, NU, VEL)
    return api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)


def test_apply_matches_face_loops(env):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(33, 20, 17)
    mat = _matrix(api, mesh, ctx, g)
    x = np.sin(0.37 * np.arange(g.n_cells))
    xv, yv = api.DeviceVector.from_numpy(ctx, x), api.DeviceVector(ctx, g.n_cells)
    mat.apply(1.0, 0.0, xv, yv)
    y_ref = oracle.StencilOperator(g, -NU, 0.0, conv=1.0, vel=VEL).apply(x)
    assert np.abs(yv.to_numpy() - y_ref).max() <= 1e-13 * np.abs(y_ref).max()


@pytest.mark.parametrize("kind,gs", [("gmres", 0), ("gmres", 1), ("bicgstab", 0)])
def test_solvers_match_oracle(env, kind, gs):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(32)
    mat = _matrix(api, mesh, ctx, g)
    b_host = np.ones(g.n_cells)
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
    s = api.GmresSolver() if kind == "gmres" else api.BiCgStabSolver()
    if kind == "gmres":
        s.num_inner_iterations, s.gram_schmidt = 30, gs
    s.record_history = True
    assert s.solve(x, b, api.HipStencilOperator(mat, 1.0, 0.0))
    ref = oracle.solve(kind, oracle.StencilOperator(g, -NU, 0.0, conv=1.0, vel=VEL), b_host, num_inner_iterations=30)
    assert ref.converged
    assert abs(s.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations))  # +-5 %
    assert np.linalg.norm(x.to_numpy() - ref.x) <= 1e-7 * np.linalg.norm(ref.x) * 50
    m = min(len(s.history), len(ref.history), 12)
    assert np.allclose(s.history[:m], ref.history[:m], rtol=1e-7)


def test_config4_full_size_properties(env):
    """128^3 (BASELINE config 4 size): the solver's residual estimate equals the true residual of the
    returned x, and the restart bookkeeping holds (applies = 1 + iterations + restarts)."""
    api, mesh, oracle, ctx = env
    n = 128
    g = mesh.structured_box(n)
    mat = _matrix(api, mesh, ctx, g)
    op = api.HipStencilOperator(mat, 1.0, 0.0)
    b = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    x = api.DeviceVector(ctx, g.n_cells)
    s = api.GmresSolver()
    s.num_inner_iterations = 30
    ok = s.solve(x, b, op)
    assert ok and s.relative_error < 1e-6
    assert s.num_applies == 1 + s.iteration + -(-s.iteration // 30)
    true_res = op.ResidualNorm(b, x)
    assert abs(true_res - s.absolute_error) <= 1e-6 * s.initial_error
    # non-symmetric: <Ax, y> != <x, Ay>
    idx = np.arange(g.n_cells)
    u = api.DeviceVector.from_numpy(ctx, np.sin(0.37 * idx))
    v = api.DeviceVector.from_numpy(ctx, np.cos(0.11 * idx))
    au, av = api.DeviceVector(ctx, g.n_cells), api.DeviceVector(ctx, g.n_cells)
    op.mul(au, u)
    op.mul(av, v)
    assert abs(api.dot_product(au, v) - api.dot_product(u, av)) > 1e-6 * abs(api.dot_product(au, v))
