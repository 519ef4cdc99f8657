"""The resident path (csrc/resident.hip): CG and BiCGStab of a lattice operator as ONE persistent kernel per solve in
which every block owns a box of the lattice -- r (BiCGStab: r, p, v) in registers, the vector the operator is applied
to in LDS, only the boxes' surfaces exchanged (self-validating granules), two / three all-reduces per iteration.  It
must be indistinguishable from the throughput path (csrc/solvers.hip) but for rounding-level differences of the dot
products: same convergence rule and counters (Solver.hpp:116-147), same iteration counts, histories, solutions -- for
every depth of the boxes, ragged runs (shorter than a line, shorter than the halo), ragged last chunks of planes,
long lines, warm starts, early exits; against the oracle too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from oracle import oracle
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    yield api, mesh, oracle, ctx
    ctx.set_option("resident_path", 1)
    ctx.set_option("resident_planes", 0)
    ctx.set_option("latency_path", 1)
    ctx.close()


def _solve(api, ctx, cls, op, b_host, resident, planes=0, x0=None, **knobs):
    ctx.set_option("resident_path", int(resident))
    ctx.set_option("resident_planes", planes)
    ctx.set_option("latency_path", 1 if resident else 0)
    s = cls()
    s.record_history = True
    for k, v in knobs.items():
        setattr(s, k, v)
    b = api.DeviceVector.from_numpy(ctx, b_host)
    x = api.DeviceVector(ctx, b_host.size) if x0 is None else api.DeviceVector.from_numpy(ctx, x0)
    before = ctx.counter("resident_solves")
    ok = s.solve(x, b, op)
    taken = ctx.counter("resident_solves") - before
    ctx.set_option("resident_path", 1)
    ctx.set_option("resident_planes", 0)
    ctx.set_option("latency_path", 1)
    return ok, s, x.to_numpy(), taken


# shape, planes per block (0: automatic).  b = nx * ny rows per plane, runs of 1024 rows:
#   (8, 8, 4): one short run; (12, 86, 5): b = 1032, the last run (8 rows) shorter than the halo (a = 12);
#   (36, 30, 8): last run 56 rows, chunks of 3, 3, 2 planes; (100, 100, 100): last run 784 rows, 4 planes per block, 250 blocks;
#   (512, 4, 3): the longest line (a = 512); (64, 64, 9) with 2 / 4 / 6 / 8 / 12 planes: ragged last chunks;
#   (128, 128, 128): 8 planes, 256 blocks -- BASELINE configs 4 and 5's size
CASES = [((8, 8, 4), 0), ((12, 86, 5), 0), ((12, 86, 5), 3), ((36, 30, 8), 3), ((24, 24, 24), 0), ((64, 64, 64), 0),
         ((100, 100, 100), 0), ((512, 4, 3), 0), ((512, 4, 3), 2), ((64, 64, 9), 2), ((64, 64, 9), 4), ((64, 64, 9), 6),
         ((64, 64, 9), 8), ((64, 64, 9), 12), ((128, 128, 128), 0)]


@pytest.mark.parametrize("shape,planes", CASES)
def test_cg_box_matches_throughput_path_and_oracle(env, shape, planes):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
    ok_r, s_r, x_r, taken = _solve(api, ctx, api.CgSolver, op, b_host, True, planes)
    ok_t, s_t, x_t, _ = _solve(api, ctx, api.CgSolver, op, b_host, False)
    assert taken == 1 and s_r.path_fallback == 0
    assert ok_r and ok_t
    assert abs(s_r.iteration - s_t.iteration) <= 1 and s_r.num_applies == s_r.iteration + 1
    m = min(len(s_r.history), len(s_t.history))
    assert np.allclose(s_r.history[:m], s_t.history[:m], rtol=1e-9)
    assert np.linalg.norm(x_r - x_t) <= 1e-9 * np.linalg.norm(x_t)
    if g.n_cells <= 64 ** 3:
        ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), b_host)
        assert abs(s_r.iteration - ref.iterations) <= max(2, int(0.02 * ref.iterations))
        assert np.linalg.norm(x_r - ref.x) <= 1e-8 * np.linalg.norm(ref.x)
    mat.close()


@pytest.mark.parametrize("shape,planes", [c for c in CASES if c[1] <= 8])
def test_bicgstab_box_matches_throughput_path_and_oracle(env, shape, planes):
    """SolverBiCgStab.hpp:60-167 on a NON-symmetric operator (convection-diffusion, BASELINE config 4's)."""
    api, mesh, oracle, ctx = env
    # (cubic cells: a box with three different spacings has more than the 32 distinct upwind weights the byte-indexed
    #  lattice records hold, and runs on the general records -- not what this file tests)
    g = mesh.structured_box(*shape, lengths=tuple(n / 64.0 for n in shape))
    wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    op = api.HipStencilOperator(mat, 1.0, 0.0)
    b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
    ok_r, s_r, x_r, taken = _solve(api, ctx, api.BiCgStabSolver, op, b_host, True, planes)
    ok_t, s_t, x_t, _ = _solve(api, ctx, api.BiCgStabSolver, op, b_host, False)
    assert taken == 1 and s_r.path_fallback == 0
    assert ok_r and ok_t and s_r.num_applies == 2 * s_r.iteration + 1
    assert abs(s_r.iteration - s_t.iteration) <= max(1, int(0.05 * s_t.iteration)), (s_r.iteration, s_t.iteration)
    m = min(len(s_r.history), len(s_t.history), 6)
    assert np.allclose(s_r.history[:m], s_t.history[:m], rtol=1e-8)
    assert np.linalg.norm(x_r - x_t) <= 1e-6 * np.linalg.norm(x_t)
    if g.n_cells <= 64 ** 3:
        ref = oracle.solve("bicgstab", oracle.StencilOperator(g, -1e-2, 0.0, conv=1.0, vel=(1.0, 0.5, 0.25)), b_host)
        assert abs(s_r.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations))
        assert np.linalg.norm(x_r - ref.x) <= 1e-6 * np.linalg.norm(ref.x)
    mat.close()


@pytest.mark.parametrize("cls_name", ["CgSolver", "BiCgStabSolver"])
def test_convergence_rule_edges_and_warm_start(env, cls_name):
    api, mesh, oracle, ctx = env
    cls = getattr(api, cls_name)
    g = mesh.structured_box(20, 18, 10, lengths=(20 / 32.0, 18 / 32.0, 10 / 32.0))  # (cubic cells: <= 32 distinct weights)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = np.ones(g.n_cells)
    # the iteration limit: not converged, exactly num_iterations iterate() calls, history of num_iterations + 1
    ok, s, _, taken = _solve(api, ctx, cls, op, b_host, True, num_iterations=9, relative_error_tolerance=0.0,
                             absolute_error_tolerance=0.0)
    assert taken == 1 and not ok and s.iteration == 9 and len(s.history) == 10
    # the initial residual already meets the absolute tolerance: no iteration, converged (Solver.hpp:124-128)
    ok, s, x, taken = _solve(api, ctx, cls, op, b_host, True, absolute_error_tolerance=1e9)
    assert taken == 1 and ok and s.iteration == 0 and not x.any()
    ok, s, x, taken = _solve(api, ctx, cls, op, b_host, True, num_iterations=0)
    assert taken == 1 and not ok and s.iteration == 0 and not x.any()
    # zero right-hand side: safe_divide keeps everything finite (Crow/MathUtils.hpp:49-52)
    ok, s, x, taken = _solve(api, ctx, cls, op, np.zeros(g.n_cells), True)
    assert taken == 1 and np.all(np.isfinite(x)) and not x.any()
    # a warm start is honoured, and a start away from the solution gives the throughput path's solve
    kind = "cg" if cls_name == "CgSolver" else "bicgstab"
    ref = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), b_host)
    ok, s, x, taken = _solve(api, ctx, cls, op, b_host, True, x0=ref.x)
    assert taken == 1 and s.initial_error <= 2e-6 * np.linalg.norm(b_host)
    x0 = 0.01 * np.cos(0.3 * np.arange(g.n_cells))
    ok_r, s_r, x_r, taken = _solve(api, ctx, cls, op, b_host, True, x0=x0)
    ok_t, s_t, x_t, _ = _solve(api, ctx, cls, op, b_host, False, x0=x0)
    assert taken == 1 and ok_r and ok_t and abs(s_r.iteration - s_t.iteration) <= 1
    assert np.isclose(s_r.initial_error, s_t.initial_error, rtol=1e-12)
    assert np.linalg.norm(x_r - x_t) <= 1e-7 * np.linalg.norm(x_t)
    ref0 = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), b_host, x0=x0)
    assert np.linalg.norm(x_r - ref0.x) <= 1e-6 * np.linalg.norm(ref0.x)
    mat.close()


def test_operators_that_are_not_halo_free_lattices_take_the_other_paths(env):
    """A renumbered box (no common offset order), a box with an odd line length, and an operator beyond the size the
    chip's registers hold run where they did before."""
    api, mesh, oracle, ctx = env
    b3 = lambda n: 1.0 + 0.25 * np.sin(0.01 * np.arange(n))  # noqa: E731
    g = mesh.structured_box(16, 16, 8)
    perm = np.random.default_rng(3).permutation(g.n_cells)
    for graph in (mesh.permute_cells(g, perm), mesh.structured_box(15, 16, 8)):
        mat = api.StencilMatrix.from_face_graph(ctx, graph)
        ok, s, x, taken = _solve(api, ctx, api.CgSolver, api.HipStencilOperator(mat, -1.0, 0.0), b3(graph.n_cells), True)
        assert ok and taken == 0
        ref = oracle.solve("cg", oracle.StencilOperator(graph, -1.0, 0.0), b3(graph.n_cells))
        assert np.linalg.norm(x - ref.x) <= 1e-8 * np.linalg.norm(ref.x)
        mat.close()
    g = mesh.structured_box(64, 64, 40, lengths=(1.0, 1.0, 40 / 64.0))
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    ctx.set_option("resident_max_rows", 100000)
    try:
        ok, s, x, taken = _solve(api, ctx, api.CgSolver, api.HipStencilOperator(mat, -1.0, 0.0), b3(g.n_cells), True)
        assert ok and taken == 0
    finally:
        ctx.set_option("resident_max_rows", 1 << 22)
        mat.close()


@pytest.mark.parametrize("kind", ["cg", "bicgstab"])
def test_resident_path_is_bitwise_reproducible_and_survives_many_solves(env, kind):
    """The sequence numbers of the exchanges and all-reduces run on from solve to solve (no buffer is ever cleared):
    200 solves on two operators of different shape in turn, then 2 000 iterations with the tolerances off three times
    -- bitwise equal histories and solutions."""
    api, mesh, oracle, ctx = env
    cls = api.CgSolver if kind == "cg" else api.BiCgStabSolver
    ga, gb = mesh.structured_box(32, 30, 12, lengths=(1.0, 30 / 32.0, 12 / 32.0)), mesh.structured_box(64, 64, 20, lengths=(1.0, 1.0, 20 / 64.0))
    ma, mb = api.StencilMatrix.from_face_graph(ctx, ga), api.StencilMatrix.from_face_graph(ctx, gb)
    opa, opb = api.HipStencilOperator(ma, -1.0, 0.0), api.HipStencilOperator(mb, -1.0, 0.0)
    ba, bb = 1.0 + 0.25 * np.sin(0.01 * np.arange(ga.n_cells)), 1.0 + 0.25 * np.cos(0.02 * np.arange(gb.n_cells))
    first = {}
    for i in range(100):
        for tag, op, bh in (("a", opa, ba), ("b", opb, bb)):
            ok, s, x, taken = _solve(api, ctx, cls, op, bh, True)
            assert ok and taken == 1
            if tag not in first:
                first[tag] = (np.array(s.history), x)
            else:
                assert np.array_equal(np.array(s.history), first[tag][0]) and np.array_equal(x, first[tag][1])
    runs = []
    for _ in range(3):
        ok, s, x, taken = _solve(api, ctx, cls, opb, bb, True, num_iterations=2000 if kind == "cg" else 150,
                                 relative_error_tolerance=0.0, absolute_error_tolerance=0.0)
        assert taken == 1
        runs.append((np.array(s.history), x))
    for h, x in runs[1:]:
        assert np.array_equal(h, runs[0][0]) and np.array_equal(x, runs[0][1])
    ma.close(), mb.close()


@pytest.mark.parametrize("shape,planes", [c for c in CASES if c[1] <= 8][:6] + [((64, 64, 64), 0), ((100, 100, 100), 0)])
def test_cg_residual_surface_published_early_is_bitwise_the_late_publish(env, shape, planes):
    """CG's exchange under its second all-reduce (option resident_early; resident.hip: res_halo MODE 2): the surface of the
    new RESIDUAL goes out before beta is known and every block forms p' = r + beta p on its halo itself -- the owner's
    expression on the owner's operands.  Histories and solutions are BITWISE those of the publish behind the all-reduce,
    solve after solve: each publish takes a fresh tag whether or not the solve goes on (the unconsumed last surface of a
    solve under a tag the next solve reuses was the defect of the first attempt) -- so short solves are interleaved."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape, lengths=tuple(s / 64.0 for s in shape))
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
    try:
        ctx.set_option("resident_early", 0)
        ok0, s0, x0, taken0 = _solve(api, ctx, api.CgSolver, op, b_host, True, planes)
        assert taken0 == 1
        ctx.set_option("resident_early", 1)
        for rep in range(5):
            ok1, s1, x1, taken1 = _solve(api, ctx, api.CgSolver, op, b_host, True, planes)
            assert taken1 == 1 and ok1 == ok0 and s1.iteration == s0.iteration
            assert np.array_equal(np.asarray(s1.history), np.asarray(s0.history)), rep
            assert np.array_equal(x1, x0), rep
            # a solve that stops after a few iterations leaves an unconsumed surface behind
            _solve(api, ctx, api.CgSolver, op, b_host + rep, True, planes, num_iterations=2 + rep)
    finally:
        ctx.set_option("resident_early", 1)
        mat.close()


@pytest.mark.parametrize("shape,planes", [c for c in CASES if c[1] <= 6 and c[0] != (128, 128, 128)] + [((64, 64, 9), 1), ((36, 30, 8), 1), ((112, 112, 30), 0), ((64, 64, 6), 6), ((32, 32, 12), 6)])
def test_bicgstab_halos_formed_from_early_surfaces_are_bitwise_the_exchanged_ones(env, shape, planes):
    """BiCGStab's early publish (resident.hip: res_bicgstab_early_kernel, boxes of at most 6 planes): the halos of
    p' = r + beta (p - omega v) and of s = r - alpha v are FORMED by every block from the halos of r, p and v it keeps -- the
    owner's expressions on the owner's operands --; what travels are the surfaces of v = A p and of the new residual,
    under the all-reduces that follow them (three waits per iteration instead of five).  Histories and solutions BITWISE
    those of the exchanged halos (option resident_early = 0), solve after solve with short solves in between (every
    publish takes a fresh tag), on a non-symmetric operator."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape, lengths=tuple(n / 64.0 for n in shape))
    wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    op = api.HipStencilOperator(mat, 1.0, 0.0)
    b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
    try:
        ctx.set_option("resident_early", 0)
        ok0, s0, x0, taken0 = _solve(api, ctx, api.BiCgStabSolver, op, b_host, True, planes)
        assert taken0 == 1 and s0.path_fallback == 0
        ctx.set_option("resident_early", 1)
        for rep in range(4):
            ok1, s1, x1, taken1 = _solve(api, ctx, api.BiCgStabSolver, op, b_host, True, planes)
            assert taken1 == 1 and s1.path_fallback == 0 and ok1 == ok0 and s1.iteration == s0.iteration
            assert np.array_equal(np.asarray(s1.history), np.asarray(s0.history)), rep
            assert np.array_equal(x1, x0), rep
            _solve(api, ctx, api.BiCgStabSolver, op, b_host + rep, True, planes, num_iterations=1 + rep)
            # ... and a CG solve in between shares the exchange buffer and the tag counter
            _solve(api, ctx, api.CgSolver, api.HipStencilOperator(mat, 1.0, 0.0), b_host, True, planes, num_iterations=3)
    finally:
        ctx.set_option("resident_early", 1)
        mat.close()


@pytest.mark.parametrize("shape,planes", [((64, 64, 9), 3), ((64, 64, 9), 4), ((64, 64, 12), 6), ((36, 30, 8), 3), ((128, 128, 32), 8), ((100, 100, 12), 4), ((64, 64, 24), 12)])
def test_coefficients_kept_from_plane_to_plane_give_the_same_bits(env, shape, planes):
    """`res_apply<TZ, CACHE>`: the fourteen coefficients of a pair of rows stay in registers from plane to plane while the weight
    words do not change (wave-uniform test) -- the byte-indexed look-ups were most of the apply's LDS reads.  Option
    resident_apply_cache = 0 decodes them in every plane: histories and solutions must agree to the bit, CG on the Poisson box
    (identical words in all interior planes) and BiCGStab on the convection-diffusion box, boxes at the lattice's top and
    bottom included (their words change between planes)."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape, lengths=tuple(n / 64.0 for n in shape))
    wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
    mats = {"cg": api.StencilMatrix.from_face_graph(ctx, g),
            "bicgstab": api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)}
    b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
    try:
        for kind, cls, alpha in (("cg", api.CgSolver, -1.0), ("bicgstab", api.BiCgStabSolver, 1.0)):
            if kind == "bicgstab" and planes > 8:  # (BiCGStab's boxes are at most 8 planes deep)
                continue
            op = api.HipStencilOperator(mats[kind], alpha, 0.0)
            runs = {}
            for cache in (0, 1):
                ctx.set_option("test_disable", 0 if cache else 4)
                ok, s, x, taken = _solve(api, ctx, cls, op, b_host, True, planes, num_iterations=60)
                assert taken == 1 and s.path_fallback == 0
                runs[cache] = (np.asarray(s.history), x)
            assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]), kind
    finally:
        ctx.set_option("test_disable", 0)
        for m in mats.values():
            m.close()


@pytest.mark.parametrize("shape,planes", [((64, 64, 9), 3), ((36, 30, 8), 3), ((64, 64, 12), 6), ((128, 128, 32), 8)])
def test_halo_formed_before_or_behind_the_update_of_the_own_rows_gives_the_same_bits(env, shape, planes):
    """CG, boxes of more than two planes (option resident_halo_interleave): behind the second all-reduce half of a block's waves
    fetch the neighbours' surfaces and form the halo of p' = r + beta p BEFORE `x += alpha p; p = r + beta p` on their own rows,
    half behind it (own rows and halo entries of the LDS copy are disjoint).  Bitwise the uniform order."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape, lengths=tuple(n / 64.0 for n in shape))
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
    try:
        runs = {}
        for il in (0, 1):
            ctx.set_option("test_disable", 0 if il else 8)
            ok, s, x, taken = _solve(api, ctx, api.CgSolver, op, b_host, True, planes)
            assert taken == 1 and s.path_fallback == 0 and ok
            runs[il] = (np.asarray(s.history), x)
        assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    finally:
        ctx.set_option("test_disable", 0)
        mat.close()
