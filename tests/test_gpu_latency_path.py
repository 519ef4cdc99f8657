"""The latency path (csrc/latency.hip): CG and BiCGStab for small operators as one cooperative persistent kernel --
vectors in registers, two (CG) / three (BiCGStab) all-reduce synchronisation points per iteration.  It must be
indistinguishable from the throughput path (csrc/solvers.hip) except for rounding-level differences of the dot products: same convergence rule and
counters (Solver.hpp:116-147), same iteration counts (+-1), same solutions, against the oracle too; for every
register variant (1 / 2 / 4 / 8 slices per wavefront, records cached in registers or re-read), ragged last slices,
rows longer than the register cache, early exits and zero iterations."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from oracle import oracle
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    yield api, mesh, oracle, ctx
    ctx.set_option("latency_path", 1)
    ctx.set_option("latency_rows", 1 << 19)
    ctx.close()


def _cg(api, ctx, op, b_host, latency, **knobs):
    ctx.set_option("latency_path", 2 if latency else 0)  # (2: the latency path itself, not the resident path of csrc/resident.hip)
    s = api.CgSolver()
    s.record_history = True
    for k, v in knobs.items():
        setattr(s, k, v)
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, b_host.size)
    ok = s.solve(x, b, op)
    ctx.set_option("latency_path", 1)
    return ok, s, x.to_numpy()


# rows: 64^3 fills 256 blocks x 16 waves with one slice each; the larger boxes need 2 / 4 / 8 slices per wavefront
@pytest.mark.parametrize("shape", [(5, 3, 2), (9, 7, 1), (24, 24, 24), (64, 64, 64), (80, 80, 80), (100, 100, 100),
                                   (128, 128, 100)])
def test_box_matches_throughput_path_and_oracle(env, shape):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape)
    ctx.set_option("latency_rows", 1 << 21)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    ctx.set_option("latency_rows", 1 << 19)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
    ok_l, s_l, x_l = _cg(api, ctx, op, b_host, True)
    ok_t, s_t, x_t = _cg(api, ctx, op, b_host, False)
    assert ok_l and ok_t
    assert abs(s_l.iteration - s_t.iteration) <= 1 and s_l.num_applies == s_l.iteration + 1
    m = min(len(s_l.history), len(s_t.history))
    assert np.allclose(s_l.history[:m], s_t.history[:m], rtol=1e-9)
    assert np.linalg.norm(x_l - x_t) <= 1e-9 * np.linalg.norm(x_t)
    if g.n_cells <= 64 ** 3:
        ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), b_host)
        assert abs(s_l.iteration - ref.iterations) <= max(2, int(0.02 * ref.iterations))
        assert np.linalg.norm(x_l - ref.x) <= 1e-8 * np.linalg.norm(ref.x)
    mat.close()


def _bicgstab(api, ctx, op, b_host, latency, cache=1, **knobs):
    ctx.set_option("latency_path", 2 if latency else 0)  # (2: the latency path itself, not the resident path of csrc/resident.hip)
    ctx.set_option("latency_cache", cache)
    s = api.BiCgStabSolver()
    s.record_history = True
    for k, v in knobs.items():
        setattr(s, k, v)
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, b_host.size)
    ok = s.solve(x, b, op)
    ctx.set_option("latency_path", 1)
    ctx.set_option("latency_cache", 1)
    return ok, s, x.to_numpy()


@pytest.mark.parametrize("shape", [(5, 3, 2), (9, 7, 1), (24, 24, 24), (64, 64, 64), (80, 80, 80), (100, 100, 100),
                                   (128, 128, 100)])
def test_bicgstab_box_matches_throughput_path_and_oracle(env, shape):
    """SolverBiCgStab.hpp:60-167 as one cooperative kernel, on a NON-symmetric operator (convection-diffusion): the
    neighbours' p and s are formed by the gathering wave from published rows; every register variant."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape)
    wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
    ctx.set_option("latency_rows", 1 << 21)
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    ctx.set_option("latency_rows", 1 << 19)
    op = api.HipStencilOperator(mat, 1.0, 0.0)
    b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
    ok_t, s_t, x_t = _bicgstab(api, ctx, op, b_host, False)
    assert ok_t
    for cache in (1, 0):
        ok_l, s_l, x_l = _bicgstab(api, ctx, op, b_host, True, cache)
        assert ok_l and s_l.num_applies == 2 * s_l.iteration + 1
        assert abs(s_l.iteration - s_t.iteration) <= max(1, int(0.05 * s_t.iteration)), (s_l.iteration, s_t.iteration)
        m = min(len(s_l.history), len(s_t.history), 6)
        assert np.allclose(s_l.history[:m], s_t.history[:m], rtol=1e-8)
        assert np.linalg.norm(x_l - x_t) <= 1e-6 * np.linalg.norm(x_t)
    if g.n_cells <= 64 ** 3:
        ref = oracle.solve("bicgstab", oracle.StencilOperator(g, -1e-2, 0.0, conv=1.0, vel=(1.0, 0.5, 0.25)), b_host)
        assert abs(s_l.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations))
        assert np.linalg.norm(x_l - ref.x) <= 1e-6 * np.linalg.norm(ref.x)
    mat.close()


def test_bicgstab_convergence_rule_edges_on_the_latency_path(env):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(10)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = np.ones(g.n_cells)
    ok, s, _ = _bicgstab(api, ctx, op, b_host, True, num_iterations=9, relative_error_tolerance=0.0,
                         absolute_error_tolerance=0.0)
    assert not ok and s.iteration == 9 and len(s.history) == 10
    ok, s, x = _bicgstab(api, ctx, op, b_host, True, absolute_error_tolerance=1e9)
    assert ok and s.iteration == 0 and not x.any()
    ok, s, x = _bicgstab(api, ctx, op, b_host, True, num_iterations=0)
    assert not ok and s.iteration == 0 and not x.any()
    ok, s, x = _bicgstab(api, ctx, op, np.zeros(g.n_cells), True)
    assert np.all(np.isfinite(x)) and not x.any()
    # a one-row operator and rows longer than the register cache
    rng = np.random.default_rng(11)
    for n, per_row in ((1, 0), (65, 3), (4099, 20)):
        if per_row:
            rows = np.repeat(np.arange(n), per_row)
            a = sp.coo_matrix((rng.random(rows.size) * 0.1, (rows, rng.integers(0, n, rows.size))), shape=(n, n)).tocsr()
            a.setdiag(0.0)
            a.eliminate_zeros()
            a = (sp.diags(np.asarray(abs(a).sum(axis=1)).ravel() + 1.0) - a).tocsr()  # non-symmetric, dominant
        else:
            a = sp.csr_matrix(np.array([[2.0]]))
        m2 = api.StencilMatrix.from_csr(ctx, a)
        bh = rng.random(n) + 0.5
        ok, s, x = _bicgstab(api, ctx, api.HipStencilOperator(m2, 1.0, 0.0), bh, True, relative_error_tolerance=1e-10,
                             absolute_error_tolerance=0.0)
        assert ok and np.linalg.norm(a @ x - bh) <= 1e-8 * np.linalg.norm(bh)
        m2.close()
    mat.close()


def test_the_reference_mesh_runs_on_the_latency_path(env):
    """step.1 (79 672 cells, 3 neighbours per row) -- the reference's own mesh, the recorded run of BASELINE.md 2."""
    import os

    api, mesh, oracle, ctx = env
    from stormruler_amd import io_tetgen

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = io_tetgen.read_triangle(os.path.join(root, "tests", "golden", "mesh", "step.1."))
    g = mesh.FaceGraph(g.n_cells, 2, g.inner, g.outer, g.area, g.center, g.volume, b_center=np.zeros((0, 2)))
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    b_host = np.sin(3 * g.center[:, 0]) * np.cos(7 * g.center[:, 1])
    ok, s, x = _cg(api, ctx, api.HipStencilOperator(mat, -1.0e-2, 1.0), b_host, True)
    assert ok and abs(s.iteration - 621) <= 12
    assert abs(np.linalg.norm(x) - 108.293490873643) <= 1e-7 * 108.293490873643
    mat.close()


def test_rows_longer_than_the_register_cache_and_ragged_sizes(env):
    api, mesh, oracle, ctx = env
    rng = np.random.default_rng(5)
    for n, per_row in ((1, 0), (63, 2), (65, 3), (1000, 12), (4099, 20)):
        if per_row:
            rows = np.repeat(np.arange(n), per_row)
            cols = rng.integers(0, n, rows.size)
            a = sp.coo_matrix((rng.random(rows.size) * 0.1, (rows, cols)), shape=(n, n)).tocsr()
            a = a + a.T
            a.setdiag(0.0)
            a.eliminate_zeros()
            a = (sp.diags(np.asarray(abs(a).sum(axis=1)).ravel() + 1.0) - a).tocsr()  # SPD, diagonally dominant
        else:
            a = sp.csr_matrix(np.array([[2.0]]))
        mat = api.StencilMatrix.from_csr(ctx, a)
        assert mat.stats()["tail_rows"] == 0
        b_host = rng.random(n) + 0.5
        op = api.HipStencilOperator(mat, 1.0, 0.0)
        ok_l, s_l, x_l = _cg(api, ctx, op, b_host, True, relative_error_tolerance=1e-10, absolute_error_tolerance=0.0)
        ok_t, s_t, x_t = _cg(api, ctx, op, b_host, False, relative_error_tolerance=1e-10, absolute_error_tolerance=0.0)
        assert ok_l and ok_t and abs(s_l.iteration - s_t.iteration) <= 1
        assert np.linalg.norm(a @ x_l - b_host) <= 1e-8 * np.linalg.norm(b_host)
        assert np.linalg.norm(x_l - x_t) <= 1e-9 * np.linalg.norm(x_t)
        mat.close()


def test_convergence_rule_edges_on_the_latency_path(env):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(10)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = np.ones(g.n_cells)
    # tolerances off: exactly num_iterations iterate() calls, converged == False
    ok, s, _ = _cg(api, ctx, op, b_host, True, num_iterations=17, relative_error_tolerance=0.0,
                   absolute_error_tolerance=0.0)
    assert not ok and s.iteration == 17 and len(s.history) == 18
    # the initial residual already meets the absolute tolerance: no iteration, converged (Solver.hpp:124-128)
    ok, s, x = _cg(api, ctx, op, b_host, True, absolute_error_tolerance=1e9)
    assert ok and s.iteration == 0 and not x.any()
    # num_iterations = 0
    ok, s, x = _cg(api, ctx, op, b_host, True, num_iterations=0)
    assert not ok and s.iteration == 0 and not x.any()
    # zero right-hand side: safe_divide keeps everything finite (Crow/MathUtils.hpp:49-52)
    ok, s, x = _cg(api, ctx, op, np.zeros(g.n_cells), True)
    assert np.all(np.isfinite(x)) and not x.any()
    # a warm start is honoured
    ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), b_host)
    ctx.set_option("latency_path", 2)
    sv = api.CgSolver()
    b, xw = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector.from_numpy(ctx, ref.x)
    sv.solve(xw, b, op)
    assert sv.initial_error <= 2e-6 * np.linalg.norm(b_host)  # started from the solution, not from zero
    ctx.set_option("latency_path", 1)
    mat.close()


@pytest.mark.parametrize("shape,m", [((7, 5, 3), 5), ((24, 24, 24), 30), ((64, 64, 64), 20), ((96, 96, 96), 30),
                                     ((128, 128, 130), 12)])
def test_cooperative_gram_schmidt_chain_matches_the_kernel_per_step_path(env, shape, m):
    """GMRES's Arnoldi orthogonalisation as one cooperative kernel (w in registers; 1 / 2 / 4 / 8 / 16 slices per
    wavefront here) against the kernel-per-basis-vector path and the oracle: same values in the same order, the
    reduction trees differ in rounding only."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape)
    wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    op = api.HipStencilOperator(mat, 1.0, 0.0)
    b_host = 1.0 + 0.25 * np.cos(0.02 * np.arange(g.n_cells))
    runs = {}
    # coop: 0 the kernel-per-step path; 1 the chain the library picks for this size; "reg" / "lds" / "quad": the chain
    # with its basis vectors through registers (two steps per synchronisation point), landing in an LDS ring by
    # LDS-DMA (two steps), or three / four steps per point with the two-level all-reduce -- every one of them forced
    for coop in (1, 0, "reg", "lds", "quad"):
        for generic in ((0, 1) if coop in (0, 1) else (0,)):  # the fused loop and the engine share the chain
            ctx.set_option("coop_mgs", 0 if coop == 0 else 1)
            ctx.set_option("coop_mgs_lds", {"reg": 0, "lds": 2, "quad": 0}.get(coop, 1))
            ctx.set_option("coop_mgs_quad", {"reg": 0, "lds": 0, "quad": 2}.get(coop, 1))
            ctx.set_option("generic_solvers", generic)
            s = api.GmresSolver()
            s.num_inner_iterations, s.record_history = m, True
            b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
            ok = s.solve(x, b, op)
            runs[(coop, generic)] = (ok, s.iteration, s.history.copy(), x.to_numpy())
    ctx.set_option("coop_mgs", 1)
    ctx.set_option("coop_mgs_lds", 1)
    ctx.set_option("coop_mgs_quad", 1)
    ctx.set_option("generic_solvers", 0)
    ok0, it0, h0, x0 = runs[(0, 0)]
    assert ok0
    for key, (ok, it, h, x) in runs.items():
        assert ok and abs(it - it0) <= 1, (key, it, it0)
        k = min(len(h), len(h0))
        assert np.allclose(h[:k], h0[:k], rtol=1e-8), key
        assert np.linalg.norm(x - x0) <= 1e-8 * np.linalg.norm(x0), key
    if g.n_cells <= 64 ** 3:
        ref = oracle.solve("gmres", oracle.StencilOperator(g, -1e-2, 0.0, conv=1.0, vel=(1.0, 0.5, 0.25)), b_host,
                           num_inner_iterations=m)
        assert abs(runs[(1, 0)][1] - ref.iterations) <= max(2, int(0.05 * ref.iterations))
        assert np.linalg.norm(runs[(1, 0)][3] - ref.x) <= 1e-7 * np.linalg.norm(ref.x)
    mat.close()


@pytest.mark.parametrize("kind", ["cg", "bicgstab"])
@pytest.mark.parametrize("shape", [(64, 64, 64), (80, 80, 80), (100, 100, 64)])
def test_latency_path_is_bitwise_reproducible(env, kind, shape):
    """Rows published by one block are gathered by others behind an all-reduce only: a row that a gathering wave
    saw too early would show as a run-to-run difference.  2 000 iterations (4 000 .. 6 000 synchronisation points,
    every block's rows republished each time) with the tolerances off, three times: bitwise equal histories."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape)
    ctx.set_option("latency_rows", 1 << 21)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    ctx.set_option("latency_rows", 1 << 19)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
    runs = []
    for _ in range(3):
        f = _cg if kind == "cg" else _bicgstab
        ok, s, x = f(api, ctx, op, b_host, True, num_iterations=2000 if kind == "cg" else 120,
                     relative_error_tolerance=0.0, absolute_error_tolerance=0.0)
        runs.append((np.array(s.history), x))
    for h, x in runs[1:]:
        assert np.array_equal(h, runs[0][0]) and np.array_equal(x, runs[0][1])
    mat.close()


@pytest.mark.parametrize("shape", [(48, 40, 36), (64, 64, 33), (20, 18, 16)])
def test_chain_kernel_applies_the_operator_itself_with_the_same_bits(shape):
    """GMRES's Arnoldi step w = A q_k (SolverGmres.hpp:155) inside the Gram-Schmidt chain kernel (test hook test_disable bit 1 switches it off;
    latency.hip: mgs_chain_quad_kernel<S, T, true>) -- spmv_canon_kernel's arithmetic on the thread's own row pairs: the
    residual histories and the solutions are BITWISE those of the launch-then-chain form, for the symmetric and the
    convection-diffusion operator; a chain variant that cannot apply gets the launch in front of it."""
    from stormruler_amd import api, mesh
    from test_gpu_convdiff import NU, VEL

    ctx = api.Context(0)
    g = mesh.structured_box(*shape, lengths=tuple(s / 64.0 for s in shape))  # (cubic cells: few distinct weights)
    wi, wo, de = mesh.convection_diffusion_weights(g, NU, VEL)
    mats = {"poisson": (api.StencilMatrix.from_face_graph(ctx, g), -1.0),
            "convdiff": (api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de), 1.0)}
    b_host = 1.0 + 0.3 * np.sin(0.02 * np.arange(g.n_cells))
    for name, (mat, alpha) in mats.items():
        assert mat.stats()["paired_rows"] == 2
        runs = {}
        for key, apply_opt, quad in (("launch", 0, 1), ("fused", 1, 1), ("register chain", 1, 0)):
            ctx.set_option("test_disable", 0 if apply_opt else 1)  # (1: the apply as a launch in front of the chain)
            ctx.set_option("coop_mgs_quad", quad)
            s = api.GmresSolver()
            s.num_inner_iterations, s.record_history, s.num_iterations = 20, True, 55
            s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
            b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
            s.solve(x, b, api.HipStencilOperator(mat, alpha, 0.0))
            assert s.path_fallback == 0
            runs[key] = (np.asarray(s.history), x.to_numpy())
        assert np.array_equal(runs["launch"][0], runs["fused"][0]) and np.array_equal(runs["launch"][1], runs["fused"][1]), name
        # (the register chain groups its sums differently from the quadruples: to rounding)
        assert np.allclose(runs["register chain"][0], runs["launch"][0], rtol=1e-9), name
        mat.close()
    ctx.close()


@pytest.mark.parametrize("shape", [(64, 64, 33), (100, 50, 40), (20, 18, 16), (128, 128, 112), (128, 128, 128)])
def test_chain_prefetch_under_the_all_reduce_changes_no_bit(shape):
    """Test hook test_disable bit 2 off / on: the next group's basis vectors requested between the block's
    arrival at the all-reduce and its wait for the others (co_allreduce_dense_arrive / _wait) -- into registers up to four
    row pairs per thread, the first vector into registers and the others through LDS (LDS-DMA) at eight (the 128^3 of
    BASELINE config 4; ragged last block at 112 planes) -- loads moved, nothing else: histories and solutions bitwise equal."""
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    g = mesh.structured_box(*shape, lengths=tuple(s / 64.0 for s in shape))
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    b_host = 1.0 + 0.3 * np.sin(0.02 * np.arange(g.n_cells))
    runs = []
    for pf in (0, 1):
        ctx.set_option("test_disable", 0 if pf else 2)  # (2: no prefetch under the all-reduce)
        s = api.GmresSolver()
        s.num_inner_iterations, s.record_history, s.num_iterations = 30, True, 75
        s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
        b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
        s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
        assert s.path_fallback == 0
        runs.append((np.asarray(s.history), x.to_numpy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    mat.close()
    ctx.close()
