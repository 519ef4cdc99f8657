"""Edge cases the domain has: empty and one-row vectors / operators, ragged sizes around the
wavefront / slice / block granularities, rows with no neighbours, isolated components, very long
rows, repeated solves on one context, vectors of mismatched contexts."""
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
    ctx.close()


def test_empty_vectors_and_operator(env):
    api, mesh, oracle, ctx = env
    a, b = api.DeviceVector(ctx, 0), api.DeviceVector(ctx, 0)
    assert api.dot_product(a, b) == 0.0 and api.norm_2(a) == 0.0
    a += 2.0 * b
    a <<= b
    api.fill_with(a, 1.0)
    assert a.to_numpy().size == 0
    mat = api.StencilMatrix.from_csr(ctx, sp.csr_matrix((0, 0)))
    mat.apply(1.0, 0.0, a, b)
    s = api.CgSolver()
    assert s.solve(a, b, api.HipStencilOperator(mat, 1.0, 0.0)) and s.iteration == 0


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1025, 2047, 2049])
def test_ragged_sizes_1d_chain(env, n):
    """1-D Laplacian chains of every awkward length: slices, blocks and the 2048-element streaming
    tiles all have ragged tails here."""
    api, mesh, oracle, ctx = env
    main = 2.0 * np.ones(n)
    a = sp.diags([-np.ones(max(n - 1, 0)), main, -np.ones(max(n - 1, 0))], [-1, 0, 1], shape=(n, n)).tocsr()
    mat = api.StencilMatrix.from_csr(ctx, a)
    x = np.sin(0.3 * np.arange(n)) + 0.1
    xv, yv = api.DeviceVector.from_numpy(ctx, x), api.DeviceVector(ctx, n)
    mat.apply(1.0, 0.0, xv, yv)
    ref = oracle.CsrOperator(a).apply(x)
    assert np.abs(yv.to_numpy() - ref).max() <= 1e-14 * max(np.abs(ref).max(), 1.0)
    s = api.CgSolver()
    s.absolute_error_tolerance, s.relative_error_tolerance = 1e-12, 0.0
    b = api.DeviceVector.from_numpy(ctx, np.ones(n))
    xs = api.DeviceVector(ctx, n)
    assert s.solve(xs, b, api.HipStencilOperator(mat, 1.0, 0.0))
    r = oracle.solve("cg", oracle.CsrOperator(a), np.ones(n), abs_tol=1e-12, rel_tol=0.0)
    assert abs(s.iteration - r.iterations) <= 2
    assert np.linalg.norm(xs.to_numpy() - r.x) <= 1e-9 * np.linalg.norm(r.x)


def test_rows_without_neighbours_and_isolated_components(env):
    api, mesh, oracle, ctx = env
    n = 300
    rng = np.random.default_rng(1)
    a = sp.lil_matrix((n, n))
    a.setdiag(3.0 + rng.random(n))
    for i in range(0, 100, 2):       # a few 2-cycles; rows 100.. are purely diagonal
        a[i, i + 1] = a[i + 1, i] = -1.0
    a = a.tocsr()
    mat = api.StencilMatrix.from_csr(ctx, a)
    st = mat.stats()
    assert st["max_row_len"] == 1
    x = rng.standard_normal(n)
    xv, yv = api.DeviceVector.from_numpy(ctx, x), api.DeviceVector(ctx, n)
    mat.apply(2.0, -1.0, xv, yv)
    assert np.allclose(yv.to_numpy(), 2.0 * (a @ x) - x, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("width", [1, 2, 3, 4, 5, 7, 8, 9, 11, 16, 17])
def test_every_slice_width(env, width):
    """Rows of exactly `width` entries exercise each case of the width dispatch (odd tails, > 8 chunks)."""
    api, mesh, oracle, ctx = env
    c2 = api.Context(0)
    c2.set_option("ell_cap", 32)
    n = 200
    rng = np.random.default_rng(width)
    rows, cols, vals = [], [], []
    for i in range(n):
        cs = rng.choice([j for j in range(n) if j != i], width, replace=False)
        rows += [i] * width
        cols += list(cs)
        vals += list(rng.standard_normal(width))
    a = (sp.coo_matrix((vals, (rows, cols)), shape=(n, n)) + sp.eye(n) * 5.0).tocsr()
    mat = api.StencilMatrix.from_csr(c2, a)
    st = mat.stats()
    assert st["max_row_len"] == width and st["tail_nnz"] == 0
    x = rng.standard_normal(n)
    xv, yv = api.DeviceVector.from_numpy(c2, x), api.DeviceVector(c2, n)
    mat.apply(1.0, 0.0, xv, yv)
    ref = oracle.CsrOperator(a).apply(x)
    assert np.abs(yv.to_numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
    mat.close()
    c2.close()


def test_repeated_solves_reuse_the_context(env):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(12)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells))
    b = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells))
    first = None
    for kind in (api.CgSolver, api.BiCgStabSolver, api.CgSolver, api.GmresSolver, api.CgSolver):
        x = api.DeviceVector(ctx, g.n_cells)
        s = kind()
        assert s.solve(x, b, op)
        if kind is api.CgSolver:
            xs = x.to_numpy()
            assert s.iteration == ref.iterations
            if first is None:
                first = xs
            assert np.array_equal(xs, first)  # bitwise reproducible run to run


def test_vectors_of_two_contexts_do_not_mix(env):
    api, mesh, oracle, ctx = env
    c2 = api.Context(0)
    a, b = api.DeviceVector(ctx, 8), api.DeviceVector(c2, 8)
    with pytest.raises(api._lib.StormHipError):
        api.dot_product(a, b)
    with pytest.raises(api._lib.StormHipError):
        a += 1.0 * b
    c2.close()


def test_no_device_memory_leak():
    """Create / solve / destroy in a loop: the device's free memory must come back (operators, work
    vectors of the three native solvers, GMRES's Hessenberg buffers, history buffers, contexts)."""
    import torch

    from stormruler_amd import api, mesh

    g = mesh.structured_box(40)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for rep in range(12):
        ctx = api.Context(0)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        op = api.HipStencilOperator(mat, -1.0, 0.0)
        b = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells))
        for cls in (api.CgSolver, api.BiCgStabSolver, api.GmresSolver):
            x = api.DeviceVector(ctx, g.n_cells)
            s = cls()
            s.record_history = True
            assert s.solve(x, b, op)
        ctx.close()  # frees every vector / operator created on it
        if rep == 1:
            torch.cuda.synchronize()
            free1, _ = torch.cuda.mem_get_info()
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert abs(free2 - free1) < 64 << 20, (free0, free1, free2)


def test_vector_storage_pool_reuse_is_invisible():
    """vec_destroy keeps the allocation for the next vec_create of the same size (context-owned pool): the new
    vector must still be value-initialised (Field::assign semantics, Field.hpp:82-84), pending kernels of the
    old owner must not leak into it, and `pool_bytes = 0` switches the pool off."""
    import torch

    from stormruler_amd import api

    ctx = api.Context(0)
    n = 1 << 20
    a = api.DeviceVector(ctx, n)
    api.fill_with(a, 7.0)
    b = api.DeviceVector(ctx, n)
    b <<= 3.0 * a        # a kernel reading `a` is still queued when `a` is released
    a._free()
    c = api.DeviceVector(ctx, n)   # reuses a's storage
    assert np.array_equal(c.to_numpy(), np.zeros(n))
    assert np.array_equal(b.to_numpy(), np.full(n, 21.0))
    # solves in a loop reuse their work vectors: free memory stays flat after the first one
    from stormruler_amd import mesh

    g = mesh.structured_box(48)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    rhs = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells))
    frees = []
    for _ in range(4):
        x = api.DeviceVector(ctx, g.n_cells)
        assert api.GmresSolver().solve(x, rhs, op)
        x._free()
        torch.cuda.synchronize()
        frees.append(torch.cuda.mem_get_info()[0])
    assert frees[1] == frees[2] == frees[3]
    ctx.set_option("pool_bytes", 0)   # trims and disables
    d = api.DeviceVector(ctx, n)
    api.fill_with(d, 1.0)
    d._free()
    e = api.DeviceVector(ctx, n)
    assert np.array_equal(e.to_numpy(), np.zeros(n))
    ctx.close()


@pytest.mark.parametrize("cls", ["CgSolver", "BiCgStabSolver", "GmresSolver", "CgsSolver"])
def test_host_poll_without_stream_markers_gives_the_same_solve(cls):
    """The host follows a solve through self-validating words the step kernels post into a pinned ring (no event behind
    every iteration: common.hpp ring_wait).  Same iteration count and bits for every lag, when the solve converges, runs
    out of iterations or has nothing to iterate."""
    from stormruler_amd import api, mesh

    g = mesh.structured_box(24)
    res = {}
    for events in (0,):
        ctx = api.Context(0)
        ctx.set_option("latency_path", 0)  # (the kernel-per-statement loops are the ones that poll)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        b = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells))
        for lag in (0, 1, 7, 63, 200):
            for iters, tol in ((2000, 1e-6), (5, 0.0), (0, 1e-6), (2000, 1e30)):
                x = api.DeviceVector(ctx, g.n_cells)
                s = getattr(api, cls)()
                s.check_lag, s.num_iterations = lag, iters
                s.absolute_error_tolerance = s.relative_error_tolerance = tol
                ok = s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
                res[(events, lag, iters, tol)] = (ok, s.iteration, s.absolute_error, x.to_numpy())
        ctx.close()
    for key, v in res.items():
        w = res[(0, 4) + key[2:]] if (0, 4) + key[2:] in res else res[(0, 7) + key[2:]]  # (the same solve with another lag)
        assert v[0] == w[0] and v[1] == w[1] and v[2] == w[2] and np.array_equal(v[3], w[3]), key
        if key[3] == 0.0:
            assert v[1] == key[2]
        if key[3] == 1e30 or key[2] == 0:
            assert v[1] == 0


def test_solve_logs_the_reference_line(caplog):
    """One INFO line per solve, the reference's `STORM_INFO("n_iter: ..., abs_err: ..., rel_err: ...")`
    (Solver.hpp:144-145), on the logger `stormruler_amd.solvers` -- from the native loop and the statement path."""
    import logging

    from stormruler_amd import api, mesh

    g = mesh.structured_box(10)
    ctx = api.Context(0)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    b = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells))
    with caplog.at_level(logging.INFO, logger="stormruler_amd.solvers"):
        for op in (api.HipStencilOperator(mat, -1.0, 0.0), api.make_operator(lambda y, x: mat.apply(-1.0, 0.0, x, y))):
            x = api.DeviceVector(ctx, g.n_cells)
            s = api.CgSolver()
            assert s.solve(x, b, op)
    lines = [r.getMessage() for r in caplog.records]
    assert len(lines) == 2 and all(ln.startswith("n_iter:") and "abs_err:" in ln and "rel_err:" in ln for ln in lines)
    ctx.close()


# ---- documented deviations, pinned (NOTES.md section 5c "Deviations") -----------------------------------------------------

def test_non_finite_input_reaches_at_most_the_stencil_offsets_in_paired_records(env):
    """Precondition of the paired record format (storm_hip.h, storm_hip_op_apply): x is finite.  With fp64 records an
    Inf in x reaches exactly the rows the reference's face loop lets it reach; in format 3 (two rows share their
    16-byte gathers; a row without a neighbour in a merged slot multiplies the gathered value by weight 0) it also
    turns rows whose row index is one stencil OFFSET away from the Inf -- but no true neighbour of it: the cells on
    the other side of a box face -- into NaN.  Nothing further away is touched."""
    api, mesh, oracle, ctx = env
    n = 12
    g = mesh.structured_box(n)
    k = (5 * n + 7) * n + 0  # a cell on the x = 0 face: row k - 1 is the LAST cell of the previous grid line
    x_host = np.sin(0.37 * np.arange(g.n_cells))
    x_host[k] = np.inf
    with np.errstate(invalid="ignore"):
        ref_bad = ~np.isfinite(oracle.StencilOperator(g, -1.0, 0.0).apply(x_host))
    assert ref_bad.sum() == 6  # the cell and its 5 true neighbours (its sixth face is a wall)
    bad = {}
    for fmt in (0, 3):
        ctx.set_option("spmv_dict", fmt)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        ctx.set_option("spmv_dict", 4)
        assert bool(mat.stats()["paired_rows"]) == (fmt == 3)
        x, y = api.DeviceVector.from_numpy(ctx, x_host), api.DeviceVector(ctx, g.n_cells)
        mat.apply(-1.0, 0.0, x, y)
        bad[fmt] = ~np.isfinite(y.to_numpy())
        mat.close()
    assert np.array_equal(bad[0], ref_bad)                      # fp64 records: the reference's propagation
    assert np.all(bad[3][ref_bad])                              # paired records: a superset ...
    extra = set(np.flatnonzero(bad[3] & ~ref_bad))
    allowed = {k - o for o in (1, -1, n, -n, n * n, -n * n)}    # ... within one stencil offset of the Inf
    assert extra and extra <= allowed, (extra, allowed)


@pytest.mark.parametrize("generic", [0, 1])
@pytest.mark.parametrize("how", ["abs_tol_above_initial_error", "num_iterations_zero"])
def test_gmres_without_a_single_iteration_leaves_x_alone(env, generic, how):
    """Deviation from SolverGmres.hpp:194-249: when NO iterate() ran (the initial residual already meets the
    absolute tolerance, Solver.hpp:124-128, or num_iterations = 0) the reference's finalize still back-substitutes
    and divides by H(0,0) = 0 -- its x comes back non-finite (the oracle reproduces that).  This build skips that
    finalize: x is returned bit for bit as it came in, on the fused loop and on the engine."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(8)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b_host, x0 = np.ones(g.n_cells), np.cos(0.1 * np.arange(g.n_cells))
    kw = {"abs_tol": 1e9} if how == "abs_tol_above_initial_error" else {"num_iterations": 0}
    with np.errstate(all="ignore"):
        ref = oracle.solve("gmres", oracle.StencilOperator(g, -1.0, 0.0), b_host, x0=x0, **kw)
    assert ref.iterations == 0 and not np.all(np.isfinite(ref.x))   # the reference's own behaviour
    s = api.GmresSolver()
    if "abs_tol" in kw:
        s.absolute_error_tolerance = 1e9
    else:
        s.num_iterations = 0
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector.from_numpy(ctx, x0)
    ctx.set_option("generic_solvers", generic)
    try:
        converged = s.solve(x, b, op)
    finally:
        ctx.set_option("generic_solvers", 0)
    assert s.iteration == 0 and converged == ("abs_tol" in kw) and converged == ref.converged
    assert abs(s.absolute_error - ref.initial_error) <= 1e-12 * ref.initial_error
    assert np.array_equal(x.to_numpy(), x0)
    mat.close()


@pytest.mark.parametrize("m", [64, 70])
def test_gmres_restart_of_64_and_more(env, m):
    """The reference takes any restart length (Solver.hpp:159); the fused loop's state slab holds restarts below 64,
    longer ones run on the general engine -- same answers (GMRES(100) used to be refused)."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(32)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    b_host = np.ones(g.n_cells)
    ref = oracle.solve("gmres", oracle.StencilOperator(g, -1.0, 0.0), b_host, num_inner_iterations=m, rel_tol=1e-10,
                       abs_tol=0.0)
    for gram_schmidt in (0, 1):
        s = api.GmresSolver()
        s.num_inner_iterations, s.gram_schmidt = m, gram_schmidt
        s.relative_error_tolerance, s.absolute_error_tolerance = 1e-10, 0.0
        b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
        assert s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0)) and ref.converged
        assert ref.iterations > m  # at least one restart happened
        assert abs(s.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations))
        assert np.linalg.norm(x.to_numpy() - ref.x) <= 1e-7 * np.linalg.norm(ref.x)
    mat.close()


def test_solve_non_uniform_shifts_an_affine_operator(env):
    """Solver.hpp:271-292: A(x) = M x + c has A(0) != 0; solve_non_uniform solves A(x) - A(0) = b - A(0) with the
    solver it is given (the shifted operator is a lambda: it runs on the engine)."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(11, 9, 7)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    c_host = 0.3 * np.cos(0.2 * np.arange(g.n_cells))
    b_host = np.ones(g.n_cells)
    cv = api.DeviceVector.from_numpy(ctx, c_host)

    def affine(y, x):
        mat.apply(-1.0, 0.0, x, y)
        y += cv

    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, g.n_cells)
    s = api.BiCgStabSolver()
    s.relative_error_tolerance, s.absolute_error_tolerance = 1e-10, 0.0
    assert api.solve_non_uniform(s, x, b, api.make_operator(affine))
    ref = oracle.solve("bicgstab", oracle.StencilOperator(g, -1.0, 0.0), b_host - c_host, rel_tol=1e-10, abs_tol=0.0)
    assert np.linalg.norm(x.to_numpy() - ref.x) <= 1e-8 * np.linalg.norm(ref.x)
    mat.close()


@pytest.mark.parametrize("kind", ["cg", "bicgstab", "cgs", "tfqmr", "idrs", "bicgstabl"])
@pytest.mark.parametrize("shape", [(9, 7, 5), (40, 40, 40), (64, 64, 64)])
def test_one_launch_reductions_give_the_same_bits(kind, shape):
    """Engine reductions finish in the partials kernel itself (option `fused_reduce`, csrc/krylov.hip
    publish_and_finish over csrc/ticket_device.hpp): run to run the residual history and x must be IDENTICAL; against
    the two-launch final pass (another folding order) they agree to rounding.  With a lambda operator and with a
    diagonal preconditioner."""
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    g = mesh.structured_box(*shape)
    wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    cls = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "cgs": api.CgsSolver, "tfqmr": api.TfqmrSolver,
           "idrs": api.IdrsSolver, "bicgstabl": api.BiCgStabLSolver}[kind]
    if kind == "cg":  # needs a symmetric operator
        mat.close()
        mat = api.StencilMatrix.from_face_graph(ctx, g)
    b = api.DeviceVector.from_numpy(ctx, 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells)))
    alpha = -1.0 if kind == "cg" else 1.0
    lam = api.make_operator(lambda y, x: mat.apply(alpha, 0.0, x, y))
    runs = {}
    for fused in (1, 11, 0):  # 11: the one-launch path a second time
        ctx.set_option("fused_reduce", min(fused, 1))
        for pre in (None, api.JacobiPreconditioner):
            api.rng_reset()
            s = cls()
            s.record_history, s.num_iterations = True, 60
            if pre is not None and kind != "cg":
                s.pre_op = pre()
                op = api.HipStencilOperator(mat, alpha, 0.0)
                ctx.set_option("generic_solvers", 1)
            else:
                op = lam
            x = api.DeviceVector(ctx, g.n_cells)
            s.solve(x, b, op)
            ctx.set_option("generic_solvers", 0)
            runs[(fused, pre is not None)] = (s.iteration, np.array(s.history), x.to_numpy())
    ctx.set_option("fused_reduce", 1)
    for with_pre in (False, True):
        a, a2, c = runs[(1, with_pre)], runs[(11, with_pre)], runs[(0, with_pre)]
        assert a[0] == a2[0] and np.array_equal(a[1], a2[1]) and np.array_equal(a[2], a2[2]), (kind, with_pre)
        k = min(len(a[1]), len(c[1]), 10)
        assert abs(a[0] - c[0]) <= 2 and np.allclose(a[1][:k], c[1][:k], rtol=1e-9), (kind, with_pre)
    mat.close()
    ctx.close()


@pytest.mark.parametrize("n_rows", [300, 2048 * 2 - 10, 2048 * 63 + 5, 2048 * 64, 2048 * 64 + 1, 2048 * 65 + 7, 2048 * 129 - 3])
def test_ticket_groups_at_their_edges(n_rows):
    """In-kernel reductions (csrc/ticket_device.hpp) with block counts around the group size of 64: one block, one
    partial group, exactly one / two groups, a last group of one block.  Fused CG and BiCGStab (latency path off)
    against the two-launch final pass: same iteration counts, solutions to rounding; the ticket path twice, bitwise."""
    from stormruler_amd import api

    ctx = api.Context(0)
    ctx.set_option("latency_path", 0)
    # a 1-D chain with a few long-range couplings, diagonally dominant (CG needs symmetry: built symmetric)
    rng = np.random.default_rng(n_rows)
    i = np.arange(n_rows - 1)
    a = sp.coo_matrix((np.full(n_rows - 1, 0.45), (i, i + 1)), shape=(n_rows, n_rows))
    j = rng.integers(0, n_rows - 7, n_rows // 3)
    a = a + sp.coo_matrix((np.full(j.size, 0.05), (j, j + 7)), shape=(n_rows, n_rows))
    a = (a + a.T).tocsr()
    a.sum_duplicates()
    m = (sp.diags(np.asarray(a.sum(axis=1)).ravel() + 0.2) - a).tocsr()
    mat = api.StencilMatrix.from_csr(ctx, m)
    b_host = 1.0 + 0.5 * np.sin(0.003 * np.arange(n_rows))
    for cls in (api.CgSolver, api.BiCgStabSolver):
        runs = []
        for ticket in (1, 1, 0):
            ctx.set_option("ticket_reduce", ticket)
            s = cls()
            s.record_history = True
            s.relative_error_tolerance, s.absolute_error_tolerance = 1e-10, 0.0
            b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, n_rows)
            assert s.solve(x, b, api.HipStencilOperator(mat, 1.0, 0.0))
            runs.append((s.iteration, np.array(s.history), x.to_numpy()))
        ctx.set_option("ticket_reduce", 1)
        assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
        assert abs(runs[0][0] - runs[2][0]) <= 1
        assert np.linalg.norm(runs[0][2] - runs[2][2]) <= 1e-9 * np.linalg.norm(runs[2][2])
        assert np.linalg.norm(m @ runs[0][2] - b_host) <= 1e-8 * np.linalg.norm(b_host)
    mat.close()
    ctx.close()


@pytest.mark.parametrize("kind", ["cg", "bicgstab", "cgs", "tfqmr", "tfqmr1", "idrs", "bicgstabl", "gmres", "fgmres"])
def test_paired_vector_statements_give_the_same_bits(kind):
    """The engine holds one vector statement back and sends two consecutive ones out as ONE pass (`lin_fuse`,
    csrc/krylov.hip lin2_kernel), or lets an independent reduction overtake it.  Elementwise statements executed per
    element in program order are the same arithmetic: histories and x IDENTICAL with the option off; lambda operator,
    Jacobi preconditioner on either side."""
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    g = mesh.structured_box(23, 17, 11)
    wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    cls = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "cgs": api.CgsSolver, "tfqmr": api.TfqmrSolver,
           "tfqmr1": api.Tfqmr1Solver, "idrs": api.IdrsSolver, "bicgstabl": api.BiCgStabLSolver,
           "gmres": api.GmresSolver, "fgmres": api.FgmresSolver}[kind]
    if kind == "cg":
        mat.close()
        mat = api.StencilMatrix.from_face_graph(ctx, g)
    alpha = -1.0 if kind == "cg" else 1.0
    b = api.DeviceVector.from_numpy(ctx, 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells)))
    lam = api.make_operator(lambda y, x: mat.apply(alpha, 0.0, x, y))
    runs = {}
    for fuse in (1, 0):
        ctx.set_option("lin_fuse", fuse)
        for variant in ("lambda", "jacobi-right", "jacobi-left"):
            if kind == "cg" and variant == "jacobi-left":
                continue  # (CG takes its preconditioner one way only)
            api.rng_reset()
            s = cls()
            s.record_history, s.num_iterations = True, 80
            op = lam
            if variant != "lambda":
                s.pre_op = api.JacobiPreconditioner()
                s.pre_side = api.PreconditionerSide.Right if variant.endswith("right") else api.PreconditionerSide.Left
                op = api.HipStencilOperator(mat, alpha, 0.0)
                ctx.set_option("generic_solvers", 1)
            x = api.DeviceVector(ctx, g.n_cells)
            s.solve(x, b, op)
            ctx.set_option("generic_solvers", 0)
            runs[(fuse, variant)] = (s.iteration, np.array(s.history), x.to_numpy())
    ctx.set_option("lin_fuse", 1)
    for (fuse, variant), r in runs.items():
        if fuse == 1:
            o = runs[(0, variant)]
            assert r[0] == o[0] and np.array_equal(r[1], o[1]) and np.array_equal(r[2], o[2]), (kind, variant)
    mat.close()
    ctx.close()
