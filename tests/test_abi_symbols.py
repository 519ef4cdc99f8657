"""The C-ABI library loads on a CPU-only box and exports every symbol include/storm_hip.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "storm_hip.h")
LIB = os.path.join(ROOT, "stormruler_amd", "libstorm_hip.so")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(storm_hip_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        import __graft_entry__ as ge

        ge.build()
    return LIB


def test_every_declared_symbol_is_exported(built):
    names = _declared()
    assert len(names) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", built], text=True)
    exported = set(re.findall(r" T (storm_hip_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in storm_hip.h but not exported: {missing}"
    stray = sorted(exported - set(names))
    assert not stray, f"exported but not declared: {stray}"


def test_binding_table_matches_header(built):
    from stormruler_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared()
    assert _lib.lib.storm_hip_abi_version() == 6


def test_library_carries_gfx950_code_only(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-n", built], capture_output=True, text=True).stdout
    blob = open(built, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_no_device_is_a_loud_error_not_a_fallback(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from stormruler_amd import api

    with pytest.raises(api._lib.StormHipError) as e:
        api.Context(0)
    assert e.value.status == -3  # STORM_HIP_E_NO_DEVICE


def test_host_side_argument_checks_do_not_need_a_device(built):
    from stormruler_amd._lib import lib

    assert lib.storm_hip_ctx_sync(None) == -1
    assert b"null" in lib.storm_hip_last_error()
    assert lib.storm_hip_vec_destroy(None) == 0 and lib.storm_hip_op_destroy(None) == 0
    # round 3's entry points validate before they touch a device too
    import ctypes as C

    h = C.c_void_p()
    assert lib.storm_hip_op_create_from_mesh(None, 1, 0, 3, 0, None, None, None, None, 0, None, None, None, None, C.byref(h)) == -1
    assert b"null" in lib.storm_hip_last_error()
    assert lib.storm_hip_vdiv(None, 1.0, None, None) == -1
    assert lib.storm_hip_multi_dot_begin(None, None, 1, None) == -1 and lib.storm_hip_multi_dot_end(None, 1, None) == -1


def test_product_code_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under stormruler_amd/ or include/ may reference it."""
    bad = []
    for base in ("stormruler_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                    t = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"\boracle\b", t, flags=re.I) and not f.endswith("mesh.py"):
                        bad.append(os.path.join(dirpath, f))
                    if "liboracle" in t or "from oracle" in t or "import oracle" in t:
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_solver_knobs_and_defaults_are_the_references():
    """Runtime configuration of the reference = public data members of the solver objects:
    same names, same defaults -- Solver.hpp:66-76,158-159, SolverBiCgStab.hpp:379-381, SolverIdrs.hpp:287-289,
    SolverRichardson.hpp:45 -- in the Python mirror, and in the C ABI's `storm_hip_solver_params_default`."""
    import ctypes as C

    from stormruler_amd import _lib, api

    for cls in (api.CgSolver, api.BiCgStabSolver, api.GmresSolver, api.FgmresSolver, api.CgsSolver, api.TfqmrSolver,
                api.Tfqmr1Solver, api.RichardsonSolver, api.BiCgStabLSolver, api.IdrsSolver, api.JfnkSolver):
        s = cls()
        assert s.num_iterations == 2000
        assert s.absolute_error_tolerance == 1.0e-6 and s.relative_error_tolerance == 1.0e-6
        assert s.pre_side == api.PreconditionerSide.Right and s.pre_op is None
        assert s.iteration == 0
    assert api.GmresSolver().num_inner_iterations == 50 and api.FgmresSolver().num_inner_iterations == 50
    assert api.BiCgStabLSolver().num_inner_iterations == 2
    assert api.IdrsSolver().num_inner_iterations == 4
    assert api.RichardsonSolver().relaxation_factor == 1.0e-4
    p = _lib.SolverParams()
    _lib.lib.storm_hip_solver_params_default(C.byref(p))
    assert (p.num_iterations, p.absolute_error_tolerance, p.relative_error_tolerance, p.num_inner_iterations) == \
        (2000, 1.0e-6, 1.0e-6, 50)
    assert p.gram_schmidt == 0  # modified Gram-Schmidt: the reference's arithmetic
