"""The C++ adapter (include/storm_hip/Storm.hpp): a driver written against the reference's
`solve<XSolver>(x, b, op)` interface, compiled by g++ and linked to libstorm_hip.so only."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "poisson_driver")


def _run(*args):
    if not os.path.exists(DRIVER):
        import __graft_entry__ as ge

        ge.build()
    out = subprocess.run([DRIVER, *map(str, args)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("kind,restart,tol", [("cg", 50, 1e-8), ("bicgstab", 50, 2e-6), ("gmres", 30, 5e-6)])
@pytest.mark.parametrize("mode", ["native", "lambda"])
def test_cpp_driver_matches_oracle(kind, restart, tol, mode):
    from oracle import oracle
    from stormruler_amd import mesh

    n = 24
    got = _run(n, kind, mode, restart)
    g = mesh.structured_box(n)
    ref = oracle.solve(kind, oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells), num_inner_iterations=restart)
    assert got["converged"] and ref.converged
    assert abs(got["iterations"] - ref.iterations) <= max(2, int(0.05 * ref.iterations))
    c = (n // 2 * n + n // 2) * n + n // 2
    assert abs(got["x_centre"] - ref.x[c]) <= tol * abs(ref.x[c]) * 10
    assert abs(got["x_norm2"] - np.linalg.norm(ref.x)) <= tol * np.linalg.norm(ref.x)
    assert got["relative_error"] < 1e-6


def test_cpp_driver_reproduces_recorded_reference_case(golden):
    # 32^3 CG -> 64 iterations, x[centre] = 5.612935319258e-02
    case = [c for c in golden["baseline_md_probe"]["cases"] if c["n"] == 32 and c["solver"] == "cg"][0]
    for mode in ("native", "lambda"):
        got = _run(32, "cg", mode)
        assert got["iterations"] == case["iterations"]
        assert abs(got["x_centre"] - case["x_centre"]) <= 1e-9 * case["x_centre"]


def test_plain_c_host_reaches_the_1d_known_answer():
    """tests/c/abi_poisson1d.c: the ABI from C99 (op_create_csr, vec_*, solve_cg with the reference's default
    knobs) on the 1-D Poisson KAT of SURVEY 8c: x[31] = 528 in 32 CG iterations."""
    exe = os.path.join(ROOT, "tests", "c", "abi_poisson1d")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["iterations"] == 32 and out["converged"] == 1 and abs(out["x31"] - 528.0) < 1e-6


def test_plain_c_host_drives_every_engine_method_through_a_callback_operator():
    """tests/c/abi_krylov_callback.c: storm_hip_krylov_* from C99 with a callback operator (what the reference's call
    site hands a solver: a lambda) -- every method x {no preconditioner, diagonal left, diagonal right} on the 1-D known
    answer, the stepping interface, and a failing callback."""
    exe = os.path.join(ROOT, "tests", "c", "abi_krylov_callback")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 9 * 3 + 3 and all(ln["ok"] for ln in lines), p.stdout + p.stderr
    unpre = {ln["method"]: ln["iterations"] for ln in lines if ln.get("side") == -1}
    assert unpre["cg"] == unpre["cgs"] == unpre["tfqmr1"] == unpre["gmres"] == 32  # the reference's counts


@pytest.mark.parametrize("kind", ["cg", "bicgstab", "gmres", "idrs"])
def test_cpp_host_loop_over_the_stepping_hooks_matches_the_device_loop(kind):
    """`solver.device_loop = false`: IterativeSolver::solve runs the reference's host loop (Solver.hpp:116-147) over
    init / iterate / finalize, which the shipped solvers implement with storm_hip_krylov_init / _iterate / _finalize."""
    outs = {}
    for mode in ("native", "stepping"):
        p = subprocess.run([DRIVER, "16", kind, mode, "20"], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        outs[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    a, b = outs["native"], outs["stepping"]
    assert a["converged"] and b["converged"] and abs(a["iterations"] - b["iterations"]) <= 1
    assert abs(a["x_norm2"] - b["x_norm2"]) <= 1e-8 * a["x_norm2"]


def test_cpp_user_defined_solver_runs_on_the_interface():
    """A solver written by a USER of the interface (tests/cpp/poisson_driver.cpp: SteepestDescentSolver derives from
    IterativeSolver and implements init / iterate with the overloaded vector statements): the base class's final
    `solve` drives it with the reference's convergence rule."""
    p = subprocess.run([DRIVER, "8", "user-steepest-descent", "native"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    import numpy as np

    from oracle import oracle
    from stormruler_amd import mesh

    g = mesh.structured_box(8)
    ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells), rel_tol=1e-12, abs_tol=0.0)
    assert out["converged"] and out["iterations"] > 20  # steepest descent needs many more steps than CG
    assert abs(out["x_norm2"] - np.linalg.norm(ref.x)) <= 1e-4 * np.linalg.norm(ref.x)


@pytest.mark.parametrize("mode", ["native", "eager", "lambda"])
def test_a_users_statement_by_statement_cg(mode):
    """`user-cg`: the reference's CgSolver body typed against Storm.hpp (SolverCg.hpp:54-126) by a user -- its host loop runs
    with the library's lazy statements (native / lambda) or with every statement a launch of its own (eager)."""
    from oracle import oracle
    from stormruler_amd import mesh

    n = 24
    got = _run(n, "user-cg", mode)
    g = mesh.structured_box(n)
    ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells))
    assert got["converged"] and abs(got["iterations"] - ref.iterations) <= 2
    assert abs(got["x_norm2"] - np.linalg.norm(ref.x)) <= 1e-8 * np.linalg.norm(ref.x)
    if mode == "eager":
        assert got["lazy_fused_dots"] == got["lazy_apply_dots"] == got["lazy_cg_steps"] == 0
    else:  # both reductions of every iteration rode in a statement's kernel (a 24^3 box is no marching-kernel lattice)
        assert got["lazy_fused_dots"] == got["iterations"] and got["lazy_apply_dots"] == got["iterations"]


def test_a_users_cg_on_a_large_lattice_runs_the_fused_cg_step():
    """104^3 (1.1 M rows: the size from which lattice operators get the tiled / marching kernels): every iteration but the
    first of the user's `iterate()` is the library's fused CG step -- the kernels of the library's own device loop, whose
    solve (checked against the oracle at 256^3 in test_gpu_full_size.py) it must reproduce."""
    n = 104
    got, dev = _run(n, "user-cg", "native"), _run(n, "cg", "native")
    assert got["converged"] and dev["converged"] and abs(got["iterations"] - dev["iterations"]) <= 1
    assert abs(got["x_norm2"] - dev["x_norm2"]) <= 1e-9 * dev["x_norm2"] and abs(got["x_centre"] - dev["x_centre"]) <= 1e-8 * abs(dev["x_centre"])
    assert got["lazy_cg_steps"] == got["iterations"] - 1 and got["lazy_apply_dots"] == 1
    assert got["lazy_fused_dots"] == got["iterations"]
