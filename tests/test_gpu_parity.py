"""Parity of the HIP path (through the C ABI) against the CPU oracle.  GPU only.

Tolerances are SURVEY.md 8d's: SpMV max|dy|/max|y| <= 1e-13; dot/norm rel <= 1e-12; CG /
BiCGStab same iteration count +-2 % (min +-2) and |x - x_ref|/|x_ref| <= 1e-8; GMRES +-5 %, 1e-7.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from stormruler_amd import api as _api

    return _api


@pytest.fixture(scope="module")
def ctx(api):
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    return o


def _mesh():
    from stormruler_amd import mesh

    return mesh


def _rel_max(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# ---- BLAS-1: the reference's own known answers, then oracle parity at size ---------------------

def test_blas1_reference_kats(api, ctx, golden):
    k = golden["unit_tests"]
    m = api.DeviceVector.from_numpy(ctx, np.array(k["real_matrix"]["mat"]))
    assert abs(api.norm_2(m) - k["real_matrix"]["norm_2"]) <= k["real_matrix"]["norm_2_eps"] * k["real_matrix"]["norm_2"]
    m1 = api.DeviceVector.from_numpy(ctx, np.array(k["dot_product"]["mat1"]))
    m2 = api.DeviceVector.from_numpy(ctx, np.array(k["dot_product"]["mat2"]))
    assert api.dot_product(m1, m2) == k["dot_product"]["dot"]
    # mat1 + 10 * (mat2 - mat3) == result, exact (BitternMath.cpp:146-151)
    e = k["expr_1"]
    a = api.DeviceVector.from_numpy(ctx, np.array(e["mat1"]))
    b = api.DeviceVector.from_numpy(ctx, np.array(e["mat2"]))
    c = api.DeviceVector.from_numpy(ctx, np.array(e["mat3"]))
    t = api.DeviceVector(ctx, 4)
    t <<= b - c
    out = api.DeviceVector(ctx, 4)
    out <<= a + e["scale"] * t
    assert np.array_equal(out.to_numpy(), np.array(e["result"]))
    # -mat1 * mat2 * 0.5 + mat3 / 0.01 == result, exact (BitternMath.cpp:153-158): elementwise product, scale,
    # true division by a scalar, sum
    e2 = k["expr_2"]
    a2 = api.DeviceVector.from_numpy(ctx, np.array(e2["mat1"]))
    b2 = api.DeviceVector.from_numpy(ctx, np.array(e2["mat2"]))
    c2 = api.DeviceVector.from_numpy(ctx, np.array(e2["mat3"]))
    prod, quot = api.DeviceVector(ctx, 4), api.DeviceVector(ctx, 4)
    api.vmul(prod, a2, b2)
    prod *= -e2["half"]                  # (-mat1 * mat2) * 0.5: exact on these values in any association
    quot <<= c2 / e2["hundredth"]
    prod += quot
    assert np.array_equal(prod.to_numpy(), np.array(e2["result"]))
    # 24 / +mat1 + (18 mat3 - 4 mat2) and 2 ((9 mat1 / mat3) - mat2), exact (BitternMath.cpp:160-171): the quotient nodes
    e3, e4 = k["expr_3"], k["expr_4"]
    a3, b3, c3 = (api.DeviceVector.from_numpy(ctx, np.array(e3[n])) for n in ("mat1", "mat2", "mat3"))
    q3, t3 = api.DeviceVector(ctx, 4), api.DeviceVector(ctx, 4)
    api.vdiv(q3, 24.0, None, a3)
    t3 <<= 18.0 * c3
    t3 -= 4.0 * b3
    q3 += t3
    assert np.array_equal(q3.to_numpy(), np.array(e3["result"]))
    q4 = api.DeviceVector(ctx, 4)
    api.vdiv(q4, 9.0, a3, c3)
    q4 -= b3
    q4 *= 2.0
    assert np.array_equal(q4.to_numpy(), np.array(e4["result"]))
    # normalize(0) = 0 -> safe_divide
    assert api.safe_divide(1.0, 0.0) == 0.0


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 257, 4099, 1 << 20, (1 << 22) + 3])
def test_blas1_matches_oracle(api, ctx, oracle, n):
    rng = np.random.default_rng(n)
    ha, hb, hc = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    a, b, c = (api.DeviceVector.from_numpy(ctx, h) for h in (ha, hb, hc))
    scale = max(1.0, np.sqrt(n))
    assert abs(api.dot_product(a, b) - oracle.dot(ha, hb)) <= 1e-12 * scale * max(1.0, abs(oracle.dot(ha, hb)))
    assert abs(api.norm_2(a) - oracle.norm2(ha)) <= 1e-12 * oracle.norm2(ha)
    # y += alpha x ; y -= alpha x ; y <<= x + beta y ; p <<= r + beta (p - omega v) ; /= ; *=
    a += 0.75 * b
    oracle.axpy(ha, 0.75, hb)
    a -= 1.25 * c
    oracle.axmy(ha, 1.25, hc)
    a <<= b + 0.5 * a
    oracle.xpay(ha, hb, 0.5)
    a <<= b + 0.3 * (a - 0.7 * c)
    oracle.bicg_p(ha, hb, 0.3, 0.7, hc)
    a /= 3.0
    oracle.div_scalar(ha, 3.0)
    a *= 1.5
    oracle.mul_scalar(ha, 1.5)
    assert _rel_max(a.to_numpy(), ha) <= 1e-14
    # r <<= b - r
    c <<= b - c
    oracle.sub_from(hc, hb)
    assert _rel_max(c.to_numpy(), hc) <= 1e-15
    api.fill_with(c, 2.5)
    assert np.all(c.to_numpy() == 2.5)


def test_multi_dot_and_multi_axpy(api, ctx, oracle):
    n, k = 100_003, 19
    rng = np.random.default_rng(7)
    ha = rng.standard_normal(n)
    hbs = [rng.standard_normal(n) for _ in range(k)]
    a = api.DeviceVector.from_numpy(ctx, ha)
    bs = [api.DeviceVector.from_numpy(ctx, h) for h in hbs]
    got = api.multi_dot(a, bs)
    want = np.array([oracle.dot(ha, h) for h in hbs])
    assert np.abs(got - want).max() <= 1e-10
    coefs = rng.standard_normal(k)
    api.multi_axpy(a, coefs, bs)
    for cf, h in zip(coefs, hbs):
        oracle.axpy(ha, cf, h)
    assert _rel_max(a.to_numpy(), ha) <= 1e-13


def api_sync(api, vs, j):
    return api.multi_dot(vs[0], vs[1 + j:2 + j])


def test_dots_in_flight(api, ctx, oracle):
    """storm_hip_multi_dot_begin / _end: eight reductions enqueued back to back, awaited out of order, give the sums
    the synchronous call gives (bit for bit: same kernel, same folding order); a ninth is refused, a request is
    good for one _end."""
    rng = np.random.default_rng(11)
    for n in (1, 777, 300_001):
        hs = [rng.standard_normal(n) for _ in range(9)]
        vs = [api.DeviceVector.from_numpy(ctx, h) for h in hs]
        sync = [api.multi_dot(vs[0], vs[1 + j:2 + j + (j % 3)]) for j in range(8)]
        pend = [api.PendingDots(vs[0], vs[1 + j:2 + j + (j % 3)]) for j in range(8)]
        with pytest.raises(api._lib.StormHipError):
            api.PendingDots(vs[0], vs[1:2])
        for j in (3, 0, 7, 1, 2, 6, 5, 4):
            got = pend[j].result()
            assert np.array_equal(got, sync[j])
            want = np.array([oracle.dot(hs[0], h) for h in hs[1 + j:2 + j + (j % 3)]])
            assert np.abs(got - want).max() <= 1e-10 * max(1.0, np.abs(want).max())
        with pytest.raises(api._lib.StormHipError):
            pend[3].result()
        # a slot freed in the MIDDLE is taken by the next request (round 3 derived the slot from the tag: eight begun,
        # the third ended, and the ninth found "no" slot free)
        pend = [api.PendingDots(vs[0], vs[1 + j:2 + j]) for j in range(8)]
        assert np.array_equal(pend[2].result(), api_sync(api, vs, 2))
        ninth = api.PendingDots(vs[0], vs[4:6])
        assert np.array_equal(ninth.result(), api.multi_dot(vs[0], vs[4:6]))
        for j in (7, 6, 5, 4, 3, 1, 0):
            assert np.array_equal(pend[j].result(), api_sync(api, vs, j))
        # the ring is free again; wide requests (k > 8) take the ordinary road inside _begin
        wide = api.PendingDots(vs[0], [vs[1 + (j % 8)] for j in range(19)])
        again = api.PendingDots(vs[0], vs[1:3])
        assert np.array_equal(again.result(), api.multi_dot(vs[0], vs[1:3]))
        assert np.array_equal(wide.result(), api.multi_dot(vs[0], [vs[1 + (j % 8)] for j in range(19)]))
    assert api.dot_product(vs[0], vs[1]) == api.multi_dot(vs[0], vs[1:2])[0]


def test_vector_semantics(api, ctx):
    v = api.DeviceVector.from_numpy(ctx, np.arange(10.0))
    w = api.DeviceVector()
    w.assign(v, True)  # assign ignores `copy` and zero-initialises (Field.hpp:82-84)
    assert w.shape() == (10, 1) and np.all(w.to_numpy() == 0.0)
    with pytest.raises(api._lib.StormHipError):
        api.dot_product(v, api.DeviceVector(ctx, 11))
    with pytest.raises(RuntimeError):
        api.make_operator(lambda y, x: None).conj_mul(v, w)


# ---- SpMV -------------------------------------------------------------------------------------

def _apply_both(api, ctx, oracle, g, alpha, beta, x):
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    xv = api.DeviceVector.from_numpy(ctx, x)
    yv = api.DeviceVector(ctx, g.n_cells)
    mat.apply(alpha, beta, xv, yv)
    return yv.to_numpy(), oracle.StencilOperator(g, alpha, beta).apply(x), mat


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 1, 1), (5, 3, 2), (16, 16, 16), (32, 32, 32), (64, 64, 64),
                                   (67, 5, 3), (128, 64, 8)])
@pytest.mark.parametrize("alpha,beta", [(-1.0, 0.0), (-1e-2, 1.0)])
def test_spmv_structured(api, ctx, oracle, shape, alpha, beta):
    g = _mesh().structured_box(*shape)
    x = np.sin(0.37 * np.arange(g.n_cells))
    y, y_ref, mat = _apply_both(api, ctx, oracle, g, alpha, beta, x)
    assert _rel_max(y, y_ref) <= 1e-13
    st = mat.stats()
    assert st["nnz_offdiag"] == 2 * g.n_faces and st["tail_nnz"] == 0


def test_spmv_smooth_input_keeps_cancellation_accuracy(api, ctx, oracle):
    # smooth x: |L x| << |x|/h^2; the difference form must not lose digits to cancellation
    mesh = _mesh()
    g = mesh.structured_box(48)
    c = g.center
    x = np.sin(np.pi * c[:, 0]) * np.sin(np.pi * c[:, 1]) * np.sin(np.pi * c[:, 2])
    y, y_ref, _ = _apply_both(api, ctx, oracle, g, -1.0, 0.0, x)
    assert _rel_max(y, y_ref) <= 1e-13


@pytest.mark.parametrize("ordering", ["random", "rcm", "tile"])
def test_spmv_permuted_mesh(api, ctx, oracle, ordering):
    mesh = _mesh()
    g0 = mesh.structured_box(24, 20, 16)
    perm = mesh.random_permutation(g0.n_cells)
    g = mesh.permute_cells(g0, perm)  # the "unstructured stress variant" of SURVEY 8d
    if ordering == "rcm":
        g = mesh.permute_cells(g, mesh.rcm_ordering(g))
    elif ordering == "tile":
        g = mesh.permute_cells(g0, mesh.tile_ordering(24, 20, 16, 4, 4))
    g.validate()
    x = np.cos(0.11 * np.arange(g.n_cells))
    y, y_ref, _ = _apply_both(api, ctx, oracle, g, -1.0, 0.0, x)
    assert _rel_max(y, y_ref) <= 1e-13


def test_spmv_csr_tail_and_hub_rows(api, ctx, oracle):
    """Rows far longer than the ELL cap spill to the wave-per-row CSR tail."""
    import scipy.sparse as sp

    rng = np.random.default_rng(3)
    n = 3000
    a = sp.random(n, n, density=0.002, random_state=5, format="lil")
    for hub in (0, 1500, 2999):  # hub rows with ~n/3 entries
        cols = rng.choice(n, n // 3, replace=False)
        a[hub, cols] = rng.standard_normal(cols.size)
    a = (a + sp.eye(n) * 3.0).tocsr()
    a.sum_duplicates()
    mat = api.StencilMatrix.from_csr(ctx, a)
    st = mat.stats()
    assert st["tail_rows"] >= 3 and st["tail_nnz"] > 0
    x = rng.standard_normal(n)
    xv, yv = api.DeviceVector.from_numpy(ctx, x), api.DeviceVector(ctx, n)
    mat.apply(1.0, 0.0, xv, yv)
    y_ref = oracle.CsrOperator(a).apply(x)
    assert _rel_max(yv.to_numpy(), y_ref) <= 1e-12
    mat.apply(-2.0, 0.5, xv, yv)
    assert _rel_max(yv.to_numpy(), 0.5 * x - 2.0 * y_ref) <= 1e-12


def test_spmv_forced_small_ell_cap(api, oracle):
    c2 = api.Context(0)
    c2.set_option("ell_cap", 3)  # 7-point rows have 6 entries: half of every row goes to the tail
    g = _mesh().structured_box(12, 10, 9)
    x = np.sin(0.37 * np.arange(g.n_cells))
    mat = api.StencilMatrix.from_face_graph(c2, g)
    assert mat.stats()["tail_nnz"] > 0
    xv, yv = api.DeviceVector.from_numpy(c2, x), api.DeviceVector(c2, g.n_cells)
    mat.apply(-1.0, 0.0, xv, yv)
    assert _rel_max(yv.to_numpy(), oracle.StencilOperator(g, -1.0, 0.0).apply(x)) <= 1e-13
    # CG through the tail path (no fused dot)
    b = api.DeviceVector.from_numpy(c2, np.ones(g.n_cells))
    xs = api.DeviceVector(c2, g.n_cells)
    s = api.CgSolver()
    assert s.solve(xs, b, api.HipStencilOperator(mat, -1.0, 0.0))
    ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells))
    assert abs(s.iteration - ref.iterations) <= 2
    assert np.linalg.norm(xs.to_numpy() - ref.x) <= 1e-8 * np.linalg.norm(ref.x)
    mat.close()
    c2.close()


@pytest.mark.parametrize("nt", [0, 1])
def test_spmv_kernel_variants_agree(api, oracle, nt):
    c2 = api.Context(0)
    c2.set_option("nontemporal", nt)
    g = _mesh().structured_box(40, 33, 17)
    x = np.sin(0.37 * np.arange(g.n_cells))
    mat = api.StencilMatrix.from_face_graph(c2, g)
    xv, yv = api.DeviceVector.from_numpy(c2, x), api.DeviceVector(c2, g.n_cells)
    mat.apply(-1.0, 0.25, xv, yv)
    assert _rel_max(yv.to_numpy(), oracle.StencilOperator(g, -1.0, 0.25).apply(x)) <= 1e-13
    mat.close()
    c2.close()


def test_bad_indices_are_rejected_on_the_host(api, ctx):
    inner = np.array([0, 1], np.int64)
    outer = np.array([1, 7], np.int64)  # cell 7 does not exist
    w = np.ones(2)
    with pytest.raises(api._lib.StormHipError):
        api.StencilMatrix.from_face_weights(ctx, 3, 0, inner, outer, w, w)


# ---- solvers ------------------------------------------------------------------------------------

def _solve_both(api, ctx, oracle, g, kind, b_host, alpha=-1.0, beta=0.0, **kw):
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, alpha, beta)
    cls = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "gmres": api.GmresSolver}[kind]
    s = cls()
    s.record_history = True
    for k_, v_ in kw.items():
        setattr(s, k_, v_)
    b = api.DeviceVector.from_numpy(ctx, b_host)
    x = api.DeviceVector(ctx, g.n_cells)
    ok = s.solve(x, b, op)
    okw = dict(num_iterations=s.num_iterations, abs_tol=s.absolute_error_tolerance, rel_tol=s.relative_error_tolerance)
    if kind == "gmres":
        okw["num_inner_iterations"] = s.num_inner_iterations
    ref = oracle.solve(kind, oracle.StencilOperator(g, alpha, beta), b_host, **okw)
    return s, ok, x.to_numpy(), ref


def _iters_close(a, b, frac):
    return abs(a - b) <= max(2, int(np.ceil(frac * b)))


@pytest.mark.parametrize("n", [8, 32, 64])
def test_cg_poisson(api, ctx, oracle, golden, n):
    g = _mesh().structured_box(n)
    s, ok, x, ref = _solve_both(api, ctx, oracle, g, "cg", np.ones(g.n_cells))
    assert ok and ref.converged
    assert _iters_close(s.iteration, ref.iterations, 0.02)
    assert s.num_applies == s.iteration + 1
    assert np.linalg.norm(x - ref.x) <= 1e-8 * np.linalg.norm(ref.x)
    # residual history follows the oracle's until rounding separates them near convergence
    m = min(len(s.history), len(ref.history), 20)
    assert np.allclose(s.history[:m], ref.history[:m], rtol=1e-9)
    # recorded reference outputs
    for case in golden["baseline_md_probe"]["cases"]:
        if case["solver"] == "cg" and case["n"] == n:
            assert s.iteration == case["iterations"]
            c = (n // 2 * n + n // 2) * n + n // 2
            assert abs(x[c] - case["x_centre"]) <= 1e-9 * abs(case["x_centre"])


def test_cg_second_rhs_and_helmholtz(api, ctx, oracle):
    mesh = _mesh()
    g = mesh.structured_box(20, 24, 28)
    c = g.center
    b = np.sin(3 * c[:, 0]) * np.cos(7 * c[:, 1]) * np.cos(2 * c[:, 2])  # SURVEY 8d RHS
    s, ok, x, ref = _solve_both(api, ctx, oracle, g, "cg", b, alpha=-1e-2, beta=1.0)
    assert ok and _iters_close(s.iteration, ref.iterations, 0.02)
    assert np.linalg.norm(x - ref.x) <= 1e-8 * np.linalg.norm(ref.x)


def test_convergence_rule_and_counters(api, ctx, oracle):
    g = _mesh().structured_box(10)
    b = np.ones(g.n_cells)
    # tolerances disabled -> exactly num_iterations iterate() calls, converged == False
    s, ok, x, ref = _solve_both(api, ctx, oracle, g, "cg", b, num_iterations=7, absolute_error_tolerance=0.0,
                                relative_error_tolerance=0.0)
    assert not ok and s.iteration == 7 == ref.iterations
    assert np.linalg.norm(x - ref.x) <= 1e-12 * np.linalg.norm(ref.x)
    # initial residual below the absolute tolerance -> zero iterations, converged
    s, ok, x, ref = _solve_both(api, ctx, oracle, g, "cg", b * 1e-9)
    assert ok and s.iteration == 0 == ref.iterations and np.all(x == 0.0)
    # relative-only
    s, ok, x, ref = _solve_both(api, ctx, oracle, g, "cg", b, absolute_error_tolerance=0.0,
                                relative_error_tolerance=1e-3)
    assert ok and s.iteration == ref.iterations and s.relative_error < 1e-3
    # zero right-hand side: safe_divide keeps everything finite (0/0 -> 0)
    s, ok, x, ref = _solve_both(api, ctx, oracle, g, "cg", b * 0.0, absolute_error_tolerance=0.0,
                                relative_error_tolerance=0.0, num_iterations=3)
    assert s.iteration == 3 and np.all(x == 0.0)
    # the lag of the host poll must not change the answer
    for lag in (1, 2, 9):
        s2, ok2, x2, _ = _solve_both(api, ctx, oracle, g, "cg", b, check_lag=lag)
        s1, ok1, x1, _ = _solve_both(api, ctx, oracle, g, "cg", b)
        assert s1.iteration == s2.iteration and np.array_equal(x1, x2)


@pytest.mark.parametrize("n", [16, 64])
def test_bicgstab_poisson(api, ctx, oracle, n):
    g = _mesh().structured_box(n)
    s, ok, x, ref = _solve_both(api, ctx, oracle, g, "bicgstab", np.ones(g.n_cells))
    assert ok and ref.converged
    assert _iters_close(s.iteration, ref.iterations, 0.05)
    assert s.num_applies == 2 * s.iteration + 1
    # both stop at rel 1e-6: solutions agree to the solve tolerance, much tighter on the history head
    assert np.linalg.norm(x - ref.x) <= 2e-6 * np.linalg.norm(ref.x)
    m = min(len(s.history), len(ref.history), 10)
    assert np.allclose(s.history[:m], ref.history[:m], rtol=1e-8)


@pytest.mark.parametrize("gs", [0, 1])
def test_gmres_poisson(api, ctx, oracle, gs):
    g = _mesh().structured_box(32)
    s, ok, x, ref = _solve_both(api, ctx, oracle, g, "gmres", np.ones(g.n_cells), num_inner_iterations=30,
                                gram_schmidt=gs)
    assert ok and ref.converged
    assert _iters_close(s.iteration, ref.iterations, 0.05)
    assert s.num_applies == ref.num_applies or gs == 1 or s.iteration != ref.iterations
    assert np.linalg.norm(x - ref.x) <= 1e-7 * np.linalg.norm(ref.x) * 50
    m = min(len(s.history), len(ref.history), 25)
    assert np.allclose(s.history[:m], ref.history[:m], rtol=1e-8)


def test_gmres_restart_edges(api, ctx, oracle):
    g = _mesh().structured_box(12)
    b = np.ones(g.n_cells)
    for m, iters in ((5, 5), (5, 6), (5, 10), (4, 7), (1, 3)):
        s, ok, x, ref = _solve_both(api, ctx, oracle, g, "gmres", b, num_inner_iterations=m, num_iterations=iters,
                                    absolute_error_tolerance=0.0, relative_error_tolerance=0.0)
        assert s.iteration == iters == ref.iterations
        assert s.num_applies == ref.num_applies
        assert np.linalg.norm(x - ref.x) <= 1e-10 * np.linalg.norm(ref.x), (m, iters)


# ---- the reference-statement path: solver templates over the BLAS-1 ABI ---------------------------

@pytest.mark.parametrize("kind", ["cg", "bicgstab", "gmres"])
def test_functional_operator_path_matches_native(api, ctx, oracle, kind):
    """`solve<XSolver>(x, b, *make_operator<Vector>(lambda))` as the playground does
    (Playground.cpp:151-167): a lambda operator forces the statement-by-statement loops."""
    g = _mesh().structured_box(14)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    lam = api.make_operator(lambda y, x: mat.apply(-1.0, 0.0, x, y))
    cls = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "gmres": api.GmresSolver}[kind]
    b_host = np.ones(g.n_cells)
    b = api.DeviceVector.from_numpy(ctx, b_host)
    x1, x2 = api.DeviceVector(ctx, g.n_cells), api.DeviceVector(ctx, g.n_cells)
    s1, s2 = cls(), cls()
    if kind == "gmres":
        s1.num_inner_iterations = s2.num_inner_iterations = 30
    assert s1.solve(x1, b, lam)
    assert s2.solve(x2, b, api.HipStencilOperator(mat, -1.0, 0.0))
    okw = {"num_inner_iterations": 30} if kind == "gmres" else {}
    ref = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), b_host, **okw)
    assert _iters_close(s1.iteration, ref.iterations, 0.05) and _iters_close(s2.iteration, ref.iterations, 0.05)
    tol = 1e-8 if kind == "cg" else 2e-6
    assert np.linalg.norm(x1.to_numpy() - ref.x) <= tol * np.linalg.norm(ref.x)
    assert np.linalg.norm(x2.to_numpy() - ref.x) <= tol * np.linalg.norm(ref.x)


def test_full_size_properties_256(api, ctx, oracle):
    """BASELINE config 2 size (256^3): too big for the oracle in seconds, so check size-independent
    properties: linearity, symmetry <Ax,y> = <x,Ay>, A applied to a constant = wall terms only,
    and a 128-row window against the oracle's assembled rows."""
    mesh = _mesh()
    n = 256
    g = mesh.structured_box(n)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    N = g.n_cells
    idx = np.arange(N)
    hx, hy = np.sin(0.37 * idx), np.cos(0.11 * idx)
    x, y = api.DeviceVector.from_numpy(ctx, hx), api.DeviceVector.from_numpy(ctx, hy)
    ax, ay, t = api.DeviceVector(ctx, N), api.DeviceVector(ctx, N), api.DeviceVector(ctx, N)
    mat.apply(-1.0, 0.0, x, ax)
    mat.apply(-1.0, 0.0, y, ay)
    # symmetry (SPD operator)
    d1, d2 = api.dot_product(ax, y), api.dot_product(x, ay)
    assert abs(d1 - d2) <= 1e-10 * abs(d1)
    # linearity: A(2x + 3y) = 2Ax + 3Ay
    t <<= x + 1.5 * y
    at = api.DeviceVector(ctx, N)
    mat.apply(-1.0, 0.0, t, at)
    ax += 1.5 * ay
    diff = at.to_numpy() - ax.to_numpy()
    assert np.abs(diff).max() <= 1e-12 * np.abs(ax.to_numpy()).max()
    # constant vector: interior rows give 0, wall rows (#walls * 2/h^2)
    one = api.DeviceVector(ctx, N)
    api.fill_with(one, 1.0)
    mat.apply(-1.0, 0.0, one, at)
    r = at.to_numpy().reshape(n, n, n)
    assert np.all(r[1:-1, 1:-1, 1:-1] == 0.0)
    assert np.isclose(r[0, 0, 0], 3 * 2.0 * n * n) and np.isclose(r[0, 5, 5], 2.0 * n * n)
    # CG at full size: the recurrence residual the solver reports equals the true residual
    # |b - A x| recomputed from x, and the A-norm of the error decreases (x^T b grows monotonically
    # for x0 = 0: phi(x) = x^T A x / 2 - x^T b is minimised over a growing Krylov space).
    b = api.DeviceVector(ctx, N)
    api.fill_with(b, 1.0)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    prev_phi = 0.0
    for iters in (5, 25):
        xs = api.DeviceVector(ctx, N)
        s = api.CgSolver()
        s.num_iterations = iters
        s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
        s.record_history = True
        assert not s.solve(xs, b, op)
        assert s.iteration == iters and s.num_applies == iters + 1
        assert np.isclose(s.history[0], np.sqrt(N), rtol=1e-14)  # |b - A 0| = sqrt(N)
        true_res = op.ResidualNorm(b, xs)
        assert abs(true_res - s.history[-1]) <= 1e-9 * s.history[0]
        op.mul(at, xs)
        phi = 0.5 * api.dot_product(xs, at) - api.dot_product(xs, b)
        assert phi < prev_phi
        prev_phi = phi
