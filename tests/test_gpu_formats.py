"""The record formats of the SpMV (fp64 weights + int32 columns; byte-indexed weights; byte-indexed weights and
column offsets; the same with two rows per lane sharing their gathers; the same with one common offset order for
the whole operator and the +-1 neighbours taken from the adjacent lanes) are lossless re-encodings: every format must give bit-identical results, the
library must pick them only when the operator qualifies, and every code path around the kernel (fused dot
products, slice lists of a partitioned operator, CSR tail, diagonal extraction, ragged last slice, non-uniform
widths) must hold for each of them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORMATS = [(0, 1), (1, 1), (1, 2), (2, 1), (2, 2), (2, 4), (3, 0), (4, 0), (5, 0)]  # (spmv_dict, spmv_spw)


@pytest.fixture(scope="module")
def env():
    from oracle import oracle
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    ctx.comm_init(api.Context.comm_unique_id(), 1, 0)  # lets the self-halo (periodic) case run
    yield api, mesh, oracle, ctx
    ctx.set_option("spmv_dict", 4)
    ctx.set_option("spmv_spw", 0)
    ctx.close()


def _build(ctx, fmt, make):
    ctx.set_option("spmv_dict", fmt[0])
    ctx.set_option("spmv_spw", fmt[1])
    m = make()
    ctx.set_option("spmv_dict", 4)
    ctx.set_option("spmv_spw", 0)
    return m


def _apply(api, ctx, mat, x, n_halo=0, alpha=-0.7, beta=0.3):
    xv = api.DeviceVector.from_numpy(ctx, x, n_halo=n_halo) if n_halo else api.DeviceVector.from_numpy(ctx, x)
    yv = api.DeviceVector(ctx, x.size, n_halo) if n_halo else api.DeviceVector(ctx, x.size)
    mat.apply(alpha, beta, xv, yv)
    return yv.to_numpy()


@pytest.mark.parametrize("shape", [(33, 20, 17), (7, 5, 3), (64, 2, 2), (3, 1, 1), (16, 10, 6), (130, 3, 2)])
def test_box_all_formats_bitwise_equal_and_match_oracle(env, shape):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(*shape)
    x = np.sin(0.37 * np.arange(g.n_cells))
    y_ref = oracle.StencilOperator(g, -0.7, 0.3).apply(x)
    ys = {}
    for fmt in FORMATS:
        mat = _build(ctx, fmt, lambda: api.StencilMatrix.from_face_graph(ctx, g))
        st = mat.stats()
        assert (st["value_dictionary_size"] > 0) == (fmt[0] >= 1)
        assert (st["offset_dictionary_size"] > 0) == (fmt[0] >= 2)
        if fmt[0] == 2:
            assert st["record_bytes"] == 1024 * st["n_slices"] and not st["paired_rows"]  # one 16-byte word per row
        if fmt[0] == 3:
            # rows always pair up when nx is even (a pair never straddles two grid lines); with an odd nx only
            # if the merged neighbour lists still fit 7 slots -- otherwise format 2 is kept
            assert st["paired_rows"] or shape[0] % 2 == 1, st
            if st["paired_rows"]:
                assert st["paired_rows"] == 1
                assert st["record_bytes"] == 1536 * st["n_slices"] and st["n_slices"] == (g.n_cells + 127) // 128
        if fmt[0] == 4 and st["paired_rows"]:
            # every box in natural ordering lists its neighbours in one common order: the canonical records
            # (no per-lane offsets: 8 B/row) are taken whenever the rows pair up at all
            assert st["paired_rows"] == 2 and st["record_bytes"] == 1024 * st["n_slices"], st
        ys[fmt] = _apply(api, ctx, mat, x)
        # the diagonal read back from every format is the same
        d = api.DeviceVector(ctx, g.n_cells)
        mat.diagonal(-0.7, 0.3, d)
        ys[fmt + ("d",)] = d.to_numpy()
        mat.close()
    for fmt in FORMATS[1:]:
        assert np.array_equal(ys[fmt], ys[FORMATS[0]]), fmt
        assert np.array_equal(ys[fmt + ("d",)], ys[FORMATS[0] + ("d",)]), fmt
    assert np.abs(ys[FORMATS[0]] - y_ref).max() <= 1e-13 * np.abs(y_ref).max()


def test_operators_that_do_not_qualify_keep_fp64_records(env):
    """Distinct weights everywhere (the reference's Triangle mesh; a box with random volumes): no dictionary;
    a permuted box keeps the value dictionary but has far too many column offsets for the second one."""
    import os

    api, mesh, oracle, ctx = env
    from stormruler_amd import io_tetgen

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tri = io_tetgen.read_triangle(os.path.join(root, "tests", "golden", "mesh", "square_nb.1."))
    box = mesh.structured_box(12)
    rng = np.random.default_rng(3)
    graded = mesh.structured_box(12)
    graded.volume = graded.volume * (0.5 + rng.random(graded.n_total))
    scrambled = mesh.permute_cells(box, mesh.random_permutation(box.n_cells))
    for g, want_v, want_o in ((tri, False, False), (graded, False, False), (scrambled, True, False), (box, True, True)):
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        st = mat.stats()
        assert (st["value_dictionary_size"] > 0) == want_v and (st["offset_dictionary_size"] > 0) == want_o
        assert bool(st["paired_rows"]) == (g is box)
        x = np.cos(0.11 * np.arange(g.n_cells))
        y = _apply(api, ctx, mat, x, alpha=-1.0, beta=0.0)
        y_ref = oracle.StencilOperator(g, -1.0, 0.0).apply(x)
        assert np.abs(y - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
        mat.close()


def test_more_than_256_distinct_values_falls_back(env):
    import scipy.sparse as sp

    api, mesh, oracle, ctx = env
    n = 4096
    for distinct, want in ((200, True), (300, False)):
        vals = 1.0 + (np.arange(n - 1) % distinct) / 1024.0
        a = sp.diags([vals, vals], [-1, 1], shape=(n, n), format="csr")
        mat = api.StencilMatrix.from_csr(ctx, a)
        # ext (row sums) adds its own distinct values; keep them few by checking what was chosen
        st = mat.stats()
        if not want:
            assert st["value_dictionary_size"] == 0
        x = np.sin(0.37 * np.arange(n))
        y = _apply(api, ctx, mat, x, alpha=1.0, beta=0.0)
        assert np.abs(y - a @ x).max() <= 1e-13 * np.abs(a @ x).max()
        mat.close()


def test_non_uniform_widths_and_wide_rows(env):
    """Value dictionary with slices of different widths (the general kernel's dictionary path), and rows with
    more than 7 neighbours (no dictionary: an index word holds 7 slots)."""
    import scipy.sparse as sp

    api, mesh, oracle, ctx = env
    n = 1000
    x = np.sin(0.37 * np.arange(n))
    # rows 0..255: 2 neighbours, rest: 4 neighbours, all weights 1 or 2
    rows, cols, vals = [], [], []
    for i in range(n):
        for o, v in ((-1, 1.0), (1, 1.0)) + (((-7, 2.0), (7, 2.0)) if i >= 256 else ()):
            if 0 <= i + o < n:
                rows.append(i), cols.append(i + o), vals.append(v)
    a = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    ys = {}
    for fmt in FORMATS:
        mat = _build(ctx, fmt, lambda: api.StencilMatrix.from_csr(ctx, a))
        ys[fmt] = _apply(api, ctx, mat, x, alpha=1.0, beta=0.0)
        if fmt[0] >= 1:
            assert mat.stats()["value_dictionary_size"] > 0
        mat.close()
    for fmt in FORMATS[1:]:
        assert np.array_equal(ys[fmt], ys[FORMATS[0]]), fmt
    assert np.abs(ys[FORMATS[0]] - a @ x).max() <= 1e-13 * np.abs(a @ x).max()
    wide = sp.diags([np.ones(n - abs(o)) for o in range(-5, 6) if o], [o for o in range(-5, 6) if o], format="csr")
    mat = api.StencilMatrix.from_csr(ctx, wide)
    assert mat.stats()["value_dictionary_size"] == 0 and mat.stats()["max_row_len"] == 10
    y = _apply(api, ctx, mat, x, alpha=1.0, beta=0.0)
    assert np.abs(y - wide @ x).max() <= 1e-13 * np.abs(wide @ x).max()
    mat.close()


@pytest.mark.parametrize("fmt", FORMATS)
def test_csr_tail_with_every_format(env, fmt):
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(10, 9, 8)
    ctx.set_option("ell_cap", 3)
    mat = _build(ctx, fmt, lambda: api.StencilMatrix.from_face_graph(ctx, g))
    ctx.set_option("ell_cap", 0)
    st = mat.stats()
    assert st["tail_rows"] > 0 and (st["value_dictionary_size"] > 0) == (fmt[0] >= 1)
    x = np.sin(0.37 * np.arange(g.n_cells))
    y_ref = oracle.StencilOperator(g, -0.7, 0.3).apply(x)
    assert np.abs(_apply(api, ctx, mat, x) - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    d = api.DeviceVector(ctx, g.n_cells)
    mat.diagonal(-0.7, 0.3, d)
    ref_d = 0.3 - 0.7 * mesh.assemble_csr(g, 1.0, 0.0).diagonal()
    assert np.abs(d.to_numpy() - ref_d).max() <= 1e-13 * np.abs(ref_d).max()
    mat.close()


@pytest.mark.parametrize("fmt", FORMATS)
def test_solvers_and_fused_dots_with_every_format(env, fmt):
    """CG / BiCGStab / GMRES on the device loop (fused <p, Ap> partials: one per wave, so their count depends on
    the slices per wave) -- iteration counts and solutions identical across formats."""
    api, mesh, oracle, ctx = env
    g = mesh.structured_box(24, 20, 18)
    mat = _build(ctx, fmt, lambda: api.StencilMatrix.from_face_graph(ctx, g))
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    ref_op = oracle.StencilOperator(g, -1.0, 0.0)
    for cls, kind in ((api.CgSolver, "cg"), (api.BiCgStabSolver, "bicgstab"), (api.GmresSolver, "gmres")):
        x = api.DeviceVector(ctx, g.n_cells)
        s = cls()
        assert s.solve(x, b, op)
        ref = oracle.solve(kind, ref_op, np.ones(g.n_cells))
        assert abs(s.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations)), (kind, s.iteration, ref.iterations)
        assert np.linalg.norm(x.to_numpy() - ref.x) <= 1e-6 * np.linalg.norm(ref.x)
    mat.close()


@pytest.mark.parametrize("fmt", FORMATS)
def test_partitioned_operator_with_every_format(env, fmt):
    """Interior / boundary slice lists + halo columns (offsets into the halo tail) through the RCCL self-exchange."""
    from test_gpu_comm import _periodic_z_local_graph

    api, mesh, oracle, ctx = env
    loc, send_idx = _periodic_z_local_graph(20, 12, 9)
    mat = _build(ctx, fmt, lambda: api.StencilMatrix.from_face_graph(ctx, loc))
    st = mat.stats()
    assert (st["offset_dictionary_size"] > 0) == (fmt[0] >= 2)
    n2p = loc.n_halo
    mat.set_halo([0], [0, n2p], send_idx, [0, n2p])
    x = np.sin(0.37 * np.arange(loc.n_cells))
    y = _apply(api, ctx, mat, x, n_halo=loc.n_halo, alpha=-1.0, beta=0.05)
    xf = np.concatenate([x, x[send_idx]])
    y_ref = oracle.StencilOperator(loc, -1.0, 0.05).apply(xf)[: loc.n_cells]
    assert np.abs(y - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    # CG through the split SpMV with fused dots
    b = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
    api.fill_with(b, 1.0)
    xs = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
    s = api.CgSolver()
    assert s.solve(xs, b, api.HipStencilOperator(mat, -1.0, 0.05))
    r = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
    api.HipStencilOperator(mat, -1.0, 0.05).Residual(r, b, xs)
    assert api.norm_2(r) <= 2e-6 * np.sqrt(loc.n_cells)
    mat.close()


@pytest.mark.parametrize("shape,expect_mixed", [((20, 12, 9), True), ((32, 16, 40), True), ((64, 64, 6), True),
                                                ((64, 64, 3), False), ((33, 7, 12), True)])
def test_mixed_records_of_a_partitioned_box(env, shape, expect_mixed):
    """A rank's slab: the groups of rows that read no halo column share one offset order (format 4), the outer
    planes keep format-3 records of their own (in boundary-list order).  Bitwise the same y as the all-format-3
    operator, the oracle's values, the same diagonal; with and without fused dots; `apply_add`."""
    from test_gpu_comm import _periodic_z_local_graph

    api, mesh, oracle, ctx = env
    loc, send_idx = _periodic_z_local_graph(*shape)
    mixed = api.StencilMatrix.from_face_graph(ctx, loc)
    ctx.set_option("spmv_mixed", 0)
    plain = api.StencilMatrix.from_face_graph(ctx, loc)
    ctx.set_option("spmv_mixed", 1)
    sm, sp_ = mixed.stats(), plain.stats()
    assert sp_["paired_rows"] in (0, 1)
    if sp_["paired_rows"] == 0:  # an odd nx whose merged neighbour lists do not fit: nothing to mix
        assert sm["paired_rows"] == 0
        return
    assert sm["paired_rows"] == (2 if expect_mixed else 1)
    assert 0 < sm["n_interior_slices"] < sm["n_slices"]
    if expect_mixed:
        n_bnd = sm["n_slices"] - sm["n_interior_slices"]
        assert sm["record_bytes"] == 1024 * sm["n_slices"] + 1536 * n_bnd
    x = np.sin(0.37 * np.arange(loc.n_cells)) + 1e-3 * np.arange(loc.n_cells)
    ys = []
    for mat in (mixed, plain):
        mat.set_halo([0], [0, loc.n_halo], send_idx, [0, loc.n_halo])
        ys.append(_apply(api, ctx, mat, x, n_halo=loc.n_halo, alpha=-1.0, beta=0.05))
    assert np.array_equal(ys[0], ys[1])
    xf = np.concatenate([x, x[send_idx]])
    y_ref = oracle.StencilOperator(loc, -1.0, 0.05).apply(xf)[: loc.n_cells]
    assert np.abs(ys[0] - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    ds = []
    for mat in (mixed, plain):
        d = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
        mat.diagonal(-1.0, 0.05, d)
        ds.append(d.to_numpy())
    assert np.array_equal(ds[0], ds[1]) and np.all(ds[0] > 0)
    # the fused-dot path (CG) and the accumulate form
    its = []
    for mat in (mixed, plain):
        b = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
        api.fill_with(b, 1.0)
        xs = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
        s = api.CgSolver()
        assert s.solve(xs, b, api.HipStencilOperator(mat, -1.0, 0.05))
        its.append((s.iteration, xs.to_numpy()))
    # (the format-4 kernel folds two groups' fused-dot terms per wavefront: <p, Ap> differs in rounding, y does not)
    assert its[0][0] == its[1][0] and np.allclose(its[0][1], its[1][1], rtol=1e-11, atol=0.0)
    us = []
    for mat in (mixed, plain):
        cv = api.DeviceVector.from_numpy(ctx, x, n_halo=loc.n_halo)
        uv = api.DeviceVector.from_numpy(ctx, np.cos(0.11 * np.arange(loc.n_cells)), n_halo=loc.n_halo)
        api.stormDivGrad(mat, uv, -1.0e-3, cv)
        us.append(uv.to_numpy())
    assert np.array_equal(us[0], us[1])
    mixed.close()
    plain.close()


@pytest.mark.parametrize("fmt", [(0, 1), (1, 2), (2, 2), (3, 0)])
@pytest.mark.parametrize("mesh_kind", ["box", "triangle", "periodic"])
def test_stormDivGrad_accumulate_form(env, fmt, mesh_kind):
    """``u += dt * div grad c`` (Playground.cpp:115-131, `storm_hip_op_apply_add`) on every record format,
    an unstructured mesh, and through the halo exchange."""
    import os

    api, mesh, oracle, ctx = env
    send_idx = None
    if mesh_kind == "box":
        g = mesh.structured_box(22, 13, 10)
    elif mesh_kind == "triangle":
        from stormruler_amd import io_tetgen

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        g = io_tetgen.read_triangle(os.path.join(root, "tests", "golden", "mesh", "square_nb.1."))
    else:
        from test_gpu_comm import _periodic_z_local_graph

        g, send_idx = _periodic_z_local_graph(20, 12, 9)
    mat = _build(ctx, fmt, lambda: api.StencilMatrix.from_face_graph(ctx, g))
    if send_idx is not None:
        mat.set_halo([0], [0, g.n_halo], send_idx, [0, g.n_halo])
    n = g.n_cells
    c = np.sin(0.37 * np.arange(n))
    u0 = np.cos(0.11 * np.arange(n))
    cv = api.DeviceVector.from_numpy(ctx, c, n_halo=g.n_halo) if g.n_halo else api.DeviceVector.from_numpy(ctx, c)
    uv = api.DeviceVector.from_numpy(ctx, u0, n_halo=g.n_halo) if g.n_halo else api.DeviceVector.from_numpy(ctx, u0)
    api.stormDivGrad(mat, uv, -1.0e-3, cv)
    api.stormDivGrad(mat, uv, 2.5e-4, cv)  # accumulates
    cf = c if send_idx is None else np.concatenate([c, c[send_idx]])
    lc = oracle.StencilOperator(g, 1.0, 0.0).apply(cf)[:n]  # M(c) through the reference-order face loops
    ref = u0 + (-1.0e-3) * lc + 2.5e-4 * lc
    assert np.abs(uv.to_numpy() - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())
    mat.close()


def test_playground_operator_lambda(env):
    """the operator the reference's only caller hands to CG (Playground.cpp:153-167), written
    statement for statement on the device interface, against the oracle's restatement of the same lambda; and
    `f <<= map(dF_dc, c)` (:142-148) from elementwise kernels, bit for bit."""
    import os

    from stormruler_amd import io_tetgen

    api, mesh, oracle, ctx = env
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = io_tetgen.read_triangle(os.path.join(root, "tests", "golden", "mesh", "square_nb.1."))
    tau, Gamma, sigma = 1.0e-3, 1.0e-4, 2.0  # Playground.cpp:113
    n = g.n_cells
    rng = np.random.default_rng(1)
    c_host, c_in_host = rng.random(n), rng.random(n)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    c, c_in = api.DeviceVector.from_numpy(ctx, c_host), api.DeviceVector.from_numpy(ctx, c_in_host)
    ones, t1, t2, f = (api.DeviceVector(ctx, n) for _ in range(4))
    api.fill_with(ones, 1.0)
    # f = 2.0 * c * (c - 1.0) * (2.0 * c - 1.0), evaluated left to right like the reference's lambda
    t1 <<= c - ones
    t2 <<= 2.0 * c + (-1.0) * ones  # 2c is exact, so this is the reference's (2.0 * c - 1.0) to the bit
    f <<= 2.0 * c
    api.vmul(f, f, t1)
    api.vmul(f, f, t2)
    assert np.array_equal(f.to_numpy(), oracle.dF_dc(c_host))
    w_hat, c_hat = api.DeviceVector(ctx, n), api.DeviceVector(ctx, n)

    def op(c_hat_out, c_in_vec):  # Playground.cpp:153-167
        nonlocal w_hat
        w_hat <<= f + sigma * (c_in_vec - c)
        api.stormDivGrad(mat, w_hat, -Gamma, c_in_vec)
        c_hat_out <<= c_in_vec
        api.stormDivGrad(mat, c_hat_out, -tau, w_hat)

    api.make_operator(op).mul(c_hat, c_in)
    ref_c_hat, ref_w_hat = oracle.ch_operator_apply(g, oracle.dF_dc(c_host), c_host, c_in_host, tau, Gamma, sigma)
    assert np.abs(w_hat.to_numpy() - ref_w_hat).max() <= 1e-13 * np.abs(ref_w_hat).max()
    assert np.abs(c_hat.to_numpy() - ref_c_hat).max() <= 1e-12 * np.abs(ref_c_hat).max()
    mat.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_structured_patterns_fuzz(env, seed):
    """Fuzz of the format selection and the pair merge (shortest common supersequence of two rows' offset lists):
    rows take random subsets of a small offset set IN RANDOM ORDER with weights from a small value set.  Whatever
    the builder picks -- pairs when every merged list fits 7 slots, one row per lane otherwise -- every format must
    reproduce the fp64-record result bit for bit, and that result the assembled matrix."""
    import scipy.sparse as sp

    api, mesh, oracle, ctx = env
    rng = np.random.default_rng(seed)
    n = int(rng.integers(130, 700))
    offsets = np.array([-9, -4, -1, 1, 2, 6, 13])
    max_per_row = int(rng.integers(2, 6))
    rp, cols, vals = [0], [], []
    for i in range(n):
        k = int(rng.integers(0, max_per_row + 1))
        cand = [o for o in rng.permutation(offsets) if 0 <= i + o < n][:k]
        if seed % 2 == 0:
            cand = sorted(cand)  # even seeds: ascending lists (the mesh-like case); odd seeds: arbitrary order
        cols += [i + int(o) for o in cand]
        vals += [float(rng.integers(1, 4)) for _ in cand]
        rp.append(len(cols))
    a = sp.csr_matrix((np.array(vals), np.array(cols, dtype=np.int64), np.array(rp, dtype=np.int64)), shape=(n, n))
    x = np.sin(0.37 * np.arange(n)) + 0.1 * rng.standard_normal(n)
    ys, kinds = {}, {}
    for fmt in [(0, 1), (1, 2), (2, 2), (3, 0)]:
        mat = _build(ctx, fmt, lambda: api.StencilMatrix.from_csr(ctx, a))
        st = mat.stats()
        kinds[fmt] = (st["value_dictionary_size"] > 0, st["offset_dictionary_size"] > 0, bool(st["paired_rows"]))
        ys[fmt] = _apply(api, ctx, mat, x, alpha=1.0, beta=0.0)
        d = api.DeviceVector(ctx, n)
        mat.diagonal(1.0, 0.0, d)
        assert np.abs(d.to_numpy() - a.diagonal()).max() <= 1e-13 * max(1.0, np.abs(a.diagonal()).max())
        mat.close()
    assert kinds[(0, 1)] == (False, False, False) and kinds[(2, 2)][:2] == (True, True)
    if seed % 2 == 0:  # ascending lists over 7 offsets always merge into <= 7 slots
        assert kinds[(3, 0)][2]
    for fmt in list(ys)[1:]:
        assert np.array_equal(ys[fmt], ys[(0, 1)]), (fmt, kinds)
    ref = a @ x
    assert np.abs(ys[(0, 1)] - ref).max() <= 1e-13 * np.abs(ref).max()


def test_from_mesh_gives_the_bits_of_host_side_coefficients(env):
    """`storm_hip_op_create_from_mesh` (the library forms A_f / d_f itself, threaded) against
    `mesh.face_coefficients` + `storm_hip_op_create_from_faces`: the same records, bit-identical y."""
    import os

    api, mesh, oracle, ctx = env
    from stormruler_amd import io_tetgen

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tri = io_tetgen.read_triangle(os.path.join(root, "tests", "golden", "mesh", "square_nb.1."))
    rng = np.random.default_rng(5)
    graded = mesh.structured_box(14, 9, 11)
    graded.volume = graded.volume * (0.5 + rng.random(graded.n_total))
    for g in (tri, graded, mesh.structured_box(33, 20, 17), mesh.structured_box(16)):
        a = api.StencilMatrix.from_face_graph(ctx, g)
        b = api.StencilMatrix.from_face_coefficients(ctx, g)
        sa, sb = a.stats(), b.stats()
        assert sa == sb
        x = np.cos(0.11 * np.arange(g.n_cells))
        assert np.array_equal(_apply(api, ctx, a, x), _apply(api, ctx, b, x))
        a.close(), b.close()


def test_the_threaded_operator_build_does_not_depend_on_the_thread_count(tmp_path):
    """Record packing runs on up to 16 host threads (rows from faces by chunks of faces, dictionaries merged in chunk
    order, records by row ranges): 1, 3 and 7 threads -- forced onto small inputs -- must give the same operator."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "build_once.py"
    script.write_text(
        "import sys, json, hashlib, os\n"
        f"sys.path.insert(0, {root!r})\n"
        "import numpy as np\n"
        "from stormruler_amd import api, mesh, io_tetgen\n"
        "ctx = api.Context(0)\n"
        "out = []\n"
        f"tri = io_tetgen.read_triangle(os.path.join({root!r}, 'tests', 'golden', 'mesh', 'square_nb.1.'))\n"
        "box = mesh.structured_box(33, 20, 17)\n"
        "scr = mesh.permute_cells(box, mesh.random_permutation(box.n_cells))\n"
        "lat = mesh.structured_box(64, 16, 12)\n"
        "for g in (tri, box, scr, lat):\n"
        "    m = api.StencilMatrix.from_face_graph(ctx, g)\n"
        "    x = api.DeviceVector.from_numpy(ctx, np.cos(0.11 * np.arange(g.n_cells)))\n"
        "    y = api.DeviceVector(ctx, g.n_cells)\n"
        "    m.apply(-0.7, 0.3, x, y)\n"
        "    st = m.stats()\n"
        "    out.append([st, hashlib.sha256(y.to_numpy().tobytes()).hexdigest()])\n"
        "print(json.dumps(out))\n")
    results = []
    for threads in ("1", "3", "7"):
        env = dict(os.environ, STORM_HIP_BUILD_THREADS=threads, STORM_HIP_BUILD_MIN_CHUNK="5")
        p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        results.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("[")][-1]))
    assert results[0] == results[1] == results[2]


def test_random_csr_operators_through_every_format_against_scipy():
    """tools/fuzz_csr.py, 80 cases: 1 ... 6 000 rows, empty to 40 entries per row, rows of up to 400 entries (CSR tail),
    empty rows, three distinct values (dictionary formats), every spmv_dict level and ELL cap: y = beta x + alpha A x to the
    rounding bound of the records' difference form.  (1 200 cases of the same generator ran clean when it was written.)"""
    from tools import fuzz_csr

    assert fuzz_csr.run(seed=7, cases=80, verbose=True) == 0
