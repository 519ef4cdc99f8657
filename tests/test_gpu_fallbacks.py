"""Fallbacks instead of errors: a cooperative (one-kernel) path that cannot be launched, or whose bounded wait gives
up, must not cost the solve -- the library runs it on the kernel-per-statement path (fresh kernels, same process) and
says so in `storm_hip_solver_result::path_fallback`.  The failures are forced with option `coop_force_fail`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from oracle import oracle
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    yield api, mesh, oracle, ctx
    ctx.set_option("coop_force_fail", 0)
    ctx.close()


def _solve(api, ctx, kind, mat, g, x0, generic=False, pre=False):
    ctx.set_option("generic_solvers", int(generic))
    s = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "gmres": api.GmresSolver}[kind]()
    if kind == "gmres":
        s.num_inner_iterations = 20
    if pre:
        s.pre_op = api.JacobiPreconditioner()
    b = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    x = api.DeviceVector.from_numpy(ctx, x0)
    ok = s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
    ctx.set_option("generic_solvers", 0)
    return ok, s, x.to_numpy()


@pytest.mark.parametrize("kind,generic,pre", [("cg", False, False), ("bicgstab", False, False), ("gmres", False, False),
                                              ("gmres", True, False), ("gmres", True, True)])
@pytest.mark.parametrize("how", [1, 2])
def test_a_failed_cooperative_path_falls_back_and_says_so(env, kind, generic, pre, how):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(24, 20, 16)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    x0 = 0.01 * np.cos(0.3 * np.arange(g.n_cells))  # a start that a botched restore of x would show
    ctx.set_option("coop_force_fail", 0)
    ok0, s0, xa = _solve(api, ctx, kind, mat, g, x0, generic, pre)
    assert ok0 and s0.path_fallback == 0
    ctx.set_option("coop_force_fail", how)
    ok1, s1, xb = _solve(api, ctx, kind, mat, g, x0, generic, pre)
    ctx.set_option("coop_force_fail", 0)
    assert ok1 and s1.path_fallback == how, (s1.path_fallback, how)
    assert abs(s1.iteration - s0.iteration) <= max(2, s0.iteration // 20)
    assert np.linalg.norm(xa - xb) <= 5e-6 * np.linalg.norm(xa)
    # and against the oracle from the same start
    ref = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells), x0=x0, num_inner_iterations=20) if not pre else None
    if ref is not None:
        assert np.linalg.norm(xb - ref.x) <= 5e-6 * np.linalg.norm(ref.x)
    mat.close()


def test_the_latency_path_is_left_when_no_register_variant_fits(env):
    """`latency_rows` raised beyond what a wavefront's registers hold (8 slices): the solve runs on the throughput path
    instead of failing (round 2: STORM_REQUIRE)."""
    api, mesh, oracle, ctx = env
    ctx.set_option("latency_rows", 1 << 23)
    ctx.set_option("latency_path", 2)  # (the latency path itself: the resident path would take this lattice)
    try:
        g = mesh.structured_box(144, 144, 128)  # 2.65 M rows: > 256 blocks x 16 waves x 8 slices x 64 rows
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        s = api.CgSolver()
        s.num_iterations = 30
        b, x = api.DeviceVector(ctx, g.n_cells), api.DeviceVector(ctx, g.n_cells)
        api.fill_with(b, 1.0)
        s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
        assert s.iteration == 30 and s.path_fallback == 1 and np.isfinite(s.absolute_error)
        mat.close()
    finally:
        ctx.set_option("latency_rows", 1 << 19)
        ctx.set_option("latency_path", 1)


@pytest.mark.parametrize("kind", ["cg", "bicgstab", "gmres"])
def test_ordinary_and_cooperative_launch_of_the_one_kernel_paths_agree(env, kind):
    """The one-kernel paths synchronise through memory; by default they are launched like any kernel (the runtime's
    cooperative launch runs on a queue of its own and costs two ~12 us gaps around every launch -- once per Arnoldi
    step for GMRES).  Test hook `test_disable` bit 32 brings hipLaunchCooperativeKernel back: the same bits either way, and
    neither is a fallback."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(24, 20, 18)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    x0 = np.zeros(g.n_cells)
    res = {}
    try:
        for plain in (1, 0):
            ctx.set_option("test_disable", 0 if plain else 32)
            ok, s, x = _solve(api, ctx, kind, mat, g, x0)
            assert ok and s.path_fallback == 0
            res[plain] = (s.iteration, s.absolute_error, x)
    finally:
        ctx.set_option("test_disable", 0)
        mat.close()
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2])
