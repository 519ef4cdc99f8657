"""The CALLER of the hot path in C++ (Playground.cpp:133-210): tests/cpp/timestep_driver.cpp, compiled
against include/storm_hip/Storm.hpp only, is run step by step against the oracle --

  * `ch`: the playground's own Cahn-Hilliard loop on the reference's Triangle mesh `square_nb.1` (`f <<= map(dF_dc, c)`,
    the warm start `c_hat <<= c`, `solve<CgSolver>(c_hat, c, *make_operator<...>(lambda))` with the lambda's two
    stormDivGrad calls, the per-step clock, `std::swap(c, c_hat)`) against `oracle.cahn_hilliard_step` (oracle_ch_apply /
    oracle_ch_dF_dc, storm_oracle.c) -- iteration counts per step equal to the oracle's +-1, the field to 1e-9;
  * `cavity`: BASELINE config 5's projection step on the same interface against the CPU restatement of
    tests/test_cavity_driver.py;

and the element map behind `f <<= map(dF_dc, c)` (`storm_hip_map`: a traced arithmetic program, each operation rounded
on its own) against numpy's evaluation of the same expressions BIT FOR BIT."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "timestep_driver")


@pytest.fixture(scope="module")
def ctx():
    from stormruler_amd import api

    c = api.Context(0)
    yield c
    c.close()


def _vec(ctx, a):
    from stormruler_amd import api

    return api.DeviceVector.from_numpy(ctx, a)


# ---- the element map --------------------------------------------------------------------------------------------------
def test_map_dF_dc_equals_the_oracle_bit_for_bit(ctx):
    """`f <<= map(dF_dc, c)` (Playground.cpp:142-148) in ONE kernel, against oracle_ch_dF_dc: same operations, same order,
    no contraction -- the same bits; odd lengths (the kernel's tail element) and a length below one block included."""
    from oracle import oracle
    from stormruler_amd import api

    rng = np.random.default_rng(7)
    for n in (1, 2, 63, 6252, 1 << 20 | 1):
        c_host = rng.random(n) * 3.0 - 1.0
        c, f = _vec(ctx, c_host), api.DeviceVector(ctx, n)
        f <<= api.map(lambda c: 2.0 * c * (c - 1.0) * (2.0 * c - 1.0), c)
        assert np.array_equal(f.to_numpy(), oracle.dF_dc(c_host)), n
        c <<= api.map(lambda c: 2.0 * c * (c - 1.0) * (2.0 * c - 1.0), c)  # in place: the operand is the target
        assert np.array_equal(c.to_numpy(), oracle.dF_dc(c_host)), n


def test_map_operations_against_numpy(ctx):
    """Every operation of the program (+ - * / neg abs sqrt min max), constants shared by bit pattern, one / two / three
    operands (the third being the target), each against numpy's evaluation of the same expression in the same order."""
    from stormruler_amd import api

    rng = np.random.default_rng(11)
    n = 100003
    a_h, b_h, y_h = rng.standard_normal(n), rng.random(n) + 0.5, rng.standard_normal(n)
    a, b = _vec(ctx, a_h), _vec(ctx, b_h)
    out = api.DeviceVector(ctx, n)
    cases = [
        (lambda a, b: (a - 0.25) / b + 3.0 * a * a, lambda a, b: (a - 0.25) / b + 3.0 * a * a),
        (lambda a, b: -(abs(a) * 0.5) + b.sqrt(), lambda a, b: -(np.abs(a) * 0.5) + np.sqrt(b)),
        (lambda a, b: a.min(b) - a.max(0.125) / 7.0, lambda a, b: np.minimum(a, b) - np.maximum(a, 0.125) / 7.0),
        (lambda a, b: 1.0 / (1.0 + a * a) - (2.0 - b) * (2.0 - b), lambda a, b: 1.0 / (1.0 + a * a) - (2.0 - b) * (2.0 - b)),
        (lambda a, b: 0.1 + (0.2 + (0.3 + (0.4 + (0.5 + (0.6 + a * b))))),  # right-nested: 8 operands at once
         lambda a, b: 0.1 + (0.2 + (0.3 + (0.4 + (0.5 + (0.6 + a * b)))))),
    ]
    for traced, ref in cases:
        out <<= api.map(traced, a, b)
        assert np.array_equal(out.to_numpy(), ref(a_h, b_h))
    # a constant alone, and one operand
    out <<= api.map(lambda a: 4.5, a)
    assert np.array_equal(out.to_numpy(), np.full(n, 4.5))
    # three operands, one of them the target: y + s * (a .* b), the nonlinear term of the cavity's predictor
    y = _vec(ctx, y_h)
    y <<= api.map(lambda y, a, b: y + (-0.125) * (a * b), y, a, b)
    assert np.array_equal(y.to_numpy(), y_h + (-0.125) * (a_h * b_h))
    y <<= api.map(lambda a, y, b: (y - a) * b, a, y, b)  # (the target in the middle)
    assert np.array_equal(y.to_numpy(), ((y_h + (-0.125) * (a_h * b_h)) - a_h) * b_h)


def test_map_rejects_what_the_kernel_cannot_run(ctx):
    from stormruler_amd import api
    from stormruler_amd._lib import StormHipError

    n = 64
    a, b, c, out = (api.DeviceVector(ctx, n) for _ in range(4))
    with pytest.raises(ValueError):  # three operands, none the target
        out <<= api.map(lambda a, b, c: a + b + c, a, b, c)
    with pytest.raises(TypeError):  # a traced element has no truth value
        api.map(lambda a: a if a else 0.0, a)
    deep = lambda a: 1.0 + (2.0 + (3.0 + (4.0 + (5.0 + (6.0 + (7.0 + (8.0 + a * a)))))))  # noqa: E731 -- 9 operands at once
    with pytest.raises(StormHipError, match="more than 8 operands"):
        out <<= api.map(deep, a)
    long = lambda a: sum((a * float(k) for k in range(1, 30)), a)  # noqa: E731 -- > 48 operations
    with pytest.raises(StormHipError, match="operations"):
        out <<= api.map(long, a)
    short = api.DeviceVector(ctx, n // 2)
    with pytest.raises(StormHipError):  # operand of another size
        out <<= api.map(lambda s: s + 1.0, short)


def _random_tree(rng, budget):
    """A random expression over the leaves 0 / 1 / 2 (operands) and constants: nested tuples (op, child...)."""
    if budget <= 1 or rng.random() < 0.25:
        return ("leaf", int(rng.integers(0, 3))) if rng.random() < 0.7 else ("const", float(rng.choice([0.5, -1.25, 3.0, 0.1, 7.0])))
    kind = rng.random()
    if kind < 0.2:
        return (str(rng.choice(["neg", "abs", "sqrtabs"])), _random_tree(rng, budget - 1))
    left = int(rng.integers(1, budget))
    # (a divisor is kept away from zero: |.| + 0.5 -- 0 / 0 and the sign of a NaN are not what is being compared)
    return (str(rng.choice(["add", "sub", "mul", "div", "min", "max"])), _random_tree(rng, left), _random_tree(rng, budget - left))


def _leaves_of(tree):
    if tree[0] == "leaf":
        return {tree[1]}
    if tree[0] == "const":
        return set()
    return set().union(*[_leaves_of(t) for t in tree[1:]])


def _evaluate(tree, leaves, traced):
    """The tree on traced scalars (api.Sym) or on numpy arrays: the same operations in the same order."""
    from stormruler_amd import api

    op = tree[0]
    if op == "leaf":
        return leaves[tree[1]]
    if op == "const":
        return tree[1]
    v = [_evaluate(t, leaves, traced) for t in tree[1:]]
    is_sym = lambda z: isinstance(z, api.Sym)  # noqa: E731
    if op == "neg":
        return -v[0]
    if op == "abs":
        return abs(v[0])
    if op == "sqrtabs":
        return abs(v[0]).sqrt() if is_sym(v[0]) else np.sqrt(np.abs(v[0]))
    if op == "add":
        return v[0] + v[1]
    if op == "sub":
        return v[0] - v[1]
    if op == "mul":
        return v[0] * v[1]
    if op == "div":
        return v[0] / (abs(v[1]) + 0.5)
    lo = op == "min"
    if is_sym(v[0]) or is_sym(v[1]):
        l = api.Sym.of(v[0])
        return l.min(v[1]) if lo else l.max(v[1])
    l, r = np.asarray(v[0], dtype=np.float64), np.asarray(v[1], dtype=np.float64)
    return np.where(r < l, r, l) if lo else np.where(l < r, r, l)  # std::min / std::max (FunctionalUtils' operands order)


def test_random_map_programs_equal_numpy_bit_for_bit(ctx):
    """120 random expressions (up to 21 leaves: every stack depth the kernel specialises for, folded and unfolded pushes,
    programs that read one, two or all three operands, the target among them or not) against numpy's evaluation of the same
    tree -- equal to the bit; lengths with and without the odd tail."""
    from stormruler_amd import api
    from stormruler_amd._lib import StormHipError

    rng = np.random.default_rng(2026)
    ran = rejected = 0
    depths = set()
    for case in range(120):
        n = int(rng.choice([1, 7, 64, 4099, 65536, 100001]))
        host = [rng.standard_normal(n), rng.random(n) * 4.0 - 1.0, rng.standard_normal(n) * 10.0]
        tree = _random_tree(rng, int(rng.integers(2, 22)))
        n_operands = int(rng.integers(1, 4))
        tree_leaves = _leaves_of(tree)
        if any(k >= n_operands for k in tree_leaves):
            n_operands = max(tree_leaves) + 1
        vecs = [_vec(ctx, h) for h in host[:n_operands]]
        # the target: one of the operands (always, with three), or a fresh vector
        target_is = int(rng.integers(0, n_operands)) if (n_operands == 3 or rng.random() < 0.4) else -1
        out = vecs[target_is] if target_is >= 0 else api.DeviceVector(ctx, n)
        fn = lambda *xs: _evaluate(tree, xs, True)  # noqa: E731
        try:
            out <<= api.map(fn, *vecs)
        except StormHipError as e:  # (a right-deep draw can need more than 8 operands at once: rejected, not wrong)
            assert any(w in str(e) for w in ("operands at once", "operations", "constants")), (case, tree, str(e))
            rejected += 1
            continue
        with np.errstate(all="ignore"):
            ref = np.broadcast_to(np.asarray(_evaluate(tree, host[:n_operands], False), dtype=np.float64), (n,))
        got = out.to_numpy()
        assert np.array_equal(got.view(np.int64), np.ascontiguousarray(ref).view(np.int64)), (case, n, tree, np.abs(got - ref).max())
        ran += 1
        depths.add(len(tree_leaves))
    assert ran >= 100 and rejected <= 20, (ran, rejected)


# ---- the time loops ---------------------------------------------------------------------------------------------------
def _run_driver(args, timeout=600, **env):
    assert os.path.exists(DRIVER), "tests/cpp/timestep_driver is not built (__graft_entry__.build() builds it)"
    p = subprocess.run([DRIVER, *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=dict(os.environ, **env))
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return lines[:-1], lines[-1]


def test_cahn_hilliard_time_loop_matches_the_oracle_step_by_step(tmp_path):
    """Playground.cpp:133-210 on the reference's `square_nb.1` (6 252 triangles): mesh read once, operator built once,
    five steps; per step the CG iteration count of the oracle's `cahn_hilliard_step` and its field to 1e-9.

    The playground's lambda is affine in c_in and goes to plain `solve` (not `solve_non_uniform`): CG never meets its
    tolerance -- the oracle runs all 2 000 iterations of every step, not converged, and so does the device (checked on
    the first step: 2 000 = 2 000, `converged` false on both sides).  Fields are compared with the solve bounded to 25
    iterations per step through the solver's public knob, where a rounding-sensitive recurrence can still be held to 1e-9."""
    from oracle import oracle
    from stormruler_amd import io_tetgen, mesh

    prefix = os.path.join(ROOT, "tests", "golden", "mesh", "square_nb.1.")
    g = io_tetgen.read_triangle(prefix)
    g = mesh.FaceGraph(g.n_cells, 2, g.inner, g.outer, g.area, g.center, g.volume, b_center=np.zeros((0, 2)))  # `interior_faces()` only
    steps = 5
    c = np.random.default_rng(2024).random(g.n_cells)  # (the reference: rand() / RAND_MAX, :181-183)
    c0_path = tmp_path / "c0.f64"
    c.tofile(c0_path)
    m = oracle.Mesh(g)
    # the playground's call as it stands: `solve<CgSolver>` with the default knobs, one step
    rows, last = _run_driver(["ch", prefix, str(c0_path), "1", str(tmp_path / "full")])
    _, res = oracle.cahn_hilliard_step(m, c)
    assert rows[0]["iterations"] == res.iterations == 2000 and not res.converged and not rows[0]["converged"]
    assert 0.1 * res.relative_error <= rows[0]["relative_error"] <= 10.0 * res.relative_error  # (the same order: no tighter claim)
    # ... and bounded to 25 iterations per step: step by step against the oracle
    cap = 25
    rows, last = _run_driver(["ch", prefix, str(c0_path), str(steps), str(tmp_path / "ch")], DRIVER_NUM_ITERATIONS=str(cap))
    assert len(rows) == steps and last["cells"] == g.n_cells and last["operator_builds"] == 1
    for k, row in enumerate(rows, 1):
        c, res = oracle.cahn_hilliard_step(m, c, num_iterations=cap)
        assert row["step"] == k and row["solves_logged"] == k
        assert row["iterations"] == res.iterations == cap and row["converged"] == res.converged
        # (read from the reference's log line, `abs_err: %-12e`: six digits)
        assert abs(row["absolute_error"] - res.absolute_error) <= 2e-6 * res.absolute_error, (k, row["absolute_error"], res.absolute_error)
        dev = np.fromfile(tmp_path / f"ch.step{k}.c.f64")
        assert dev.shape == c.shape and np.abs(dev - c).max() <= 1e-9 * np.abs(c).max(), (k, np.abs(dev - c).max())
        assert row["seconds"] > 0
    assert all(r["iterations"] > 0 for r in rows) and abs(last["total_time"] - sum(r["seconds"] for r in rows)) <= 1e-5


def test_cahn_hilliard_time_loop_through_solve_non_uniform_converges_like_the_oracle(tmp_path):
    """The playground's loop with the one call changed that its affine lambda calls for -- `solve_non_uniform`
    (Solver.hpp:271-292: z = A(0), f = b - z, the operator y = A(x) - z) -- converges: 50 - 56 CG iterations per step on
    `square_nb.1`.  Per step: the oracle's iteration count +-1, the residual the solve stopped at to the log line's six
    digits when the counts agree, the field to 1e-8 (two solves that each stop at rel 1e-6 of a residual of ~1e-2)."""
    from oracle import oracle
    from stormruler_amd import io_tetgen, mesh

    prefix = os.path.join(ROOT, "tests", "golden", "mesh", "square_nb.1.")
    g = io_tetgen.read_triangle(prefix)
    g = mesh.FaceGraph(g.n_cells, 2, g.inner, g.outer, g.area, g.center, g.volume, b_center=np.zeros((0, 2)))
    steps = 6
    c = np.random.default_rng(2024).random(g.n_cells)
    c0_path = tmp_path / "c0.f64"
    c.tofile(c0_path)
    rows, last = _run_driver(["ch-nonuniform", prefix, str(c0_path), str(steps), str(tmp_path / "chn")])
    assert len(rows) == steps and last["operator_builds"] == 1
    m = oracle.Mesh(g)
    for k, row in enumerate(rows, 1):
        c, res = oracle.cahn_hilliard_step_non_uniform(m, c)
        assert res.converged and row["converged"] and 20 <= res.iterations <= 200, (k, res.iterations)
        assert abs(row["iterations"] - res.iterations) <= 1, (k, row["iterations"], res.iterations)
        if row["iterations"] == res.iterations:
            assert abs(row["absolute_error"] - res.absolute_error) <= 1e-4 * res.absolute_error, (k, row["absolute_error"], res.absolute_error)
        dev = np.fromfile(tmp_path / f"chn.step{k}.c.f64")
        assert np.abs(dev - c).max() <= 1e-8 * np.abs(c).max(), (k, np.abs(dev - c).max())
        c = dev  # (the next step starts from the DEVICE's field on both sides: the comparison is per step, not accumulated)


def test_cavity_time_loop_matches_the_cpu_restatement_step_by_step(tmp_path):
    """BASELINE config 5's caller in C++: the projection step of stormruler_amd/cavity.py typed against Storm.hpp, eight
    operators built once, a warm-started pressure-Poisson CG per step -- against tests/test_cavity_driver.CpuCavity."""
    from test_cavity_driver import CpuCavity

    n, nu, steps = 16, 0.05, 6
    rows, last = _run_driver(["cavity", str(n), repr(nu), str(steps), str(tmp_path / "cav")])
    assert len(rows) == steps and last["cells"] == n ** 3 and last["operator_builds"] == 8
    cpu = CpuCavity(n, nu, last["dt"])
    assert last["dt"] == 0.2 * min(1.0 / n, (1.0 / n) ** 2 / (6.0 * nu))
    for k, row in enumerate(rows, 1):
        it_c, ok_c = cpu.step()
        assert ok_c and row["converged"]
        # (a singular -- pure Neumann -- system: the tail of the iteration is rounding-sensitive, as in test_cavity_driver)
        assert abs(row["iterations"] - it_c) <= max(3, int(0.1 * it_c)), (k, row["iterations"], it_c)
        scale = max(np.abs(cpu.u[0]).max(), 1e-30)
        for name, ref in (("ux", cpu.u[0]), ("uy", cpu.u[1]), ("uz", cpu.u[2])):
            dev = np.fromfile(tmp_path / f"cav.step{k}.{name}.f64")
            assert np.abs(dev - ref).max() <= 1e-7 * scale, (k, name)
        pd, pc = np.fromfile(tmp_path / f"cav.step{k}.p.f64"), cpu.p  # defined up to a constant: the mean-free parts
        assert np.abs((pd - pd.mean()) - (pc - pc.mean())).max() <= 1e-6 * max(np.abs(pc - pc.mean()).max(), 1e-30)
    assert all(r["seconds"] > 0 for r in rows) and abs(last["total_time"] - sum(r["seconds"] for r in rows)) <= 1e-5
