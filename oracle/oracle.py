"""ctypes binding of the CPU oracle (oracle/storm_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by anything under stormruler_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATHS = {"strict": os.path.join(_HERE, "liboracle.so"),
              "fma": os.path.join(_HERE, "liboracle_fma.so"),
              # investigation builds (make variants): reductions summed pairwise / in one long double
              "pairwise": os.path.join(_HERE, "liboracle_pairwise.so"),
              "longdouble": os.path.join(_HERE, "liboracle_longdouble.so"),
              "devlike": os.path.join(_HERE, "liboracle_devlike.so")}
_INVESTIGATION = ("pairwise", "longdouble", "devlike")
# SURVEY.md 8d's CPU-baseline flags, built ON THE MACHINE THAT RUNS THEM (never in-tree: -march=native code does not
# travel): bench.py's cpu_baseline reports these beside the strict build, which stays the parity checker.
_NATIVE_FLAGS = {"native": ["-O3", "-march=native"],
                 "native_fast": ["-O3", "-march=native", "-ffast-math"]}  # (the reference's Release: -Ofast -march=native, CMakeLists.txt:194-195)

f64p = C.POINTER(C.c_double)
i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> None:
    """Compile both oracle variants (strict: -ffp-contract=off; fma: contraction allowed)."""
    src = os.path.join(_HERE, "storm_oracle.c")
    stale = [p for k, p in _LIB_PATHS.items() if k not in _INVESTIGATION
             if force or not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src)]
    omp_so, omp_src = os.path.join(_HERE, "liboracle_omp.so"), os.path.join(_HERE, "storm_oracle_omp.c")
    if force or not os.path.exists(omp_so) or os.path.getmtime(omp_so) < os.path.getmtime(omp_src):
        stale.append(omp_so)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


class _Mesh(C.Structure):
    _fields_ = [("n_cells", C.c_int64), ("n_faces", C.c_int64), ("n_bfaces", C.c_int64),
                ("dim", C.c_int32),
                ("inner", i64p), ("outer", i64p), ("area", f64p), ("center", f64p),
                ("volume", f64p), ("b_cell", i64p), ("b_area", f64p), ("b_center", f64p),
                ("b_ghost", f64p)]


class _StencilOp(C.Structure):
    _fields_ = [("mesh", C.POINTER(_Mesh)), ("alpha", C.c_double), ("beta", C.c_double),
                ("conv", C.c_double), ("vel", C.c_double * 3)]


class _CsrOp(C.Structure):
    _fields_ = [("n", C.c_int64), ("row_ptr", i64p), ("col", i64p), ("val", f64p)]


class _Params(C.Structure):
    _fields_ = [("num_iterations", C.c_int64), ("absolute_error_tolerance", C.c_double),
                ("relative_error_tolerance", C.c_double), ("num_inner_iterations", C.c_int64),
                ("relaxation_factor", C.c_double)]


class _Result(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("absolute_error", C.c_double),
                ("relative_error", C.c_double), ("initial_error", C.c_double),
                ("converged", C.c_int32), ("num_applies", C.c_int64)]


APPLY_FN = C.CFUNCTYPE(None, C.c_void_p, f64p, f64p)

_libs = {}


def build_native(variant: str) -> str:
    """`gcc -O3 -march=native [-ffast-math]` of storm_oracle.c into a scratch directory of THIS machine; returns the path.
    Raises when there is no compiler."""
    import shutil
    import tempfile

    cc = shutil.which(os.environ.get("CC", "gcc")) or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler on this machine")
    out = os.path.join(tempfile.gettempdir(), f"liboracle_{variant}_{os.getuid()}_{os.getpid()}.so")
    subprocess.check_call([cc, *_NATIVE_FLAGS[variant], "-fPIC", "-fvisibility=hidden", "-std=c11", "-shared", "-o", out,
                           os.path.join(_HERE, "storm_oracle.c"), "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    _LIB_PATHS[variant] = out
    import atexit

    atexit.register(lambda path=out: os.path.exists(path) and os.remove(path))  # (scratch of this process only)
    return out


def native_flags(variant: str) -> str:
    return "gcc " + " ".join(_NATIVE_FLAGS[variant])


def lib(variant: str = "strict"):
    if variant not in _libs:
        build()
        if variant in _NATIVE_FLAGS and variant not in _LIB_PATHS:
            build_native(variant)
        if variant in _INVESTIGATION and not os.path.exists(_LIB_PATHS[variant]):
            subprocess.check_call(["make", "-C", _HERE, "variants"], stdout=subprocess.DEVNULL)
        L = _libs[variant] = C.CDLL(_LIB_PATHS[variant])
        L.oracle_safe_divide.restype = C.c_double
        L.oracle_safe_divide.argtypes = [C.c_double, C.c_double]
        L.oracle_sym_ortho.argtypes = [C.c_double, C.c_double, f64p, f64p, f64p]
        for name in ("oracle_norm2", "oracle_sum", "oracle_norm1", "oracle_norm_inf"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_int64, f64p]
        L.oracle_dot.restype = C.c_double
        L.oracle_dot.argtypes = [C.c_int64, f64p, f64p]
        L.oracle_copy.argtypes = [C.c_int64, f64p, f64p]
        L.oracle_fill.argtypes = [C.c_int64, f64p, C.c_double]
        L.oracle_axpy.argtypes = [C.c_int64, f64p, C.c_double, f64p]
        L.oracle_axmy.argtypes = [C.c_int64, f64p, C.c_double, f64p]
        L.oracle_xpay.argtypes = [C.c_int64, f64p, f64p, C.c_double]
        L.oracle_bicg_p.argtypes = [C.c_int64, f64p, f64p, C.c_double, C.c_double, f64p]
        L.oracle_sub_from.argtypes = [C.c_int64, f64p, f64p]
        L.oracle_div_scalar.argtypes = [C.c_int64, f64p, C.c_double]
        L.oracle_mul_scalar.argtypes = [C.c_int64, f64p, C.c_double]
        L.oracle_expr1.argtypes = [C.c_int64, f64p, f64p, C.c_double, f64p, f64p]
        L.oracle_vdiv.argtypes = [C.c_int64, f64p, C.c_double, f64p, f64p]
        L.oracle_rng_next.restype = C.c_uint64
        L.oracle_fill_randomly.argtypes = [C.c_int64, f64p]
        L.oracle_divgrad.argtypes = [C.POINTER(_Mesh), f64p, C.c_double, f64p]
        L.oracle_convection.argtypes = [C.POINTER(_Mesh), f64p, C.c_double, f64p, f64p]
        L.oracle_stencil_apply.argtypes = [C.c_void_p, f64p, f64p]
        L.oracle_csr_apply.argtypes = [C.c_void_p, f64p, f64p]
        L.oracle_gather_apply.argtypes = [C.c_void_p, f64p, f64p]
        for name in ("oracle_solve_cg", "oracle_solve_bicgstab", "oracle_solve_gmres", "oracle_solve_richardson",
                     "oracle_solve_cgs", "oracle_solve_tfqmr", "oracle_solve_tfqmr1", "oracle_solve_bicgstabl",
                     "oracle_solve_idrs"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int64, f64p, f64p,
                                         C.POINTER(_Params), C.POINTER(_Result), f64p]
        L.oracle_set_preconditioner.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_last_pre_applies.restype = C.c_int64
        L.oracle_solve_jfnk.restype = C.c_int64
        L.oracle_solve_jfnk.argtypes = L.oracle_solve_cg.argtypes
        L.oracle_solve_gmres_pre.restype = C.c_int64
        L.oracle_solve_gmres_pre.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.c_int64, f64p, f64p, C.POINTER(_Params), C.POINTER(_Result), f64p]
        L.oracle_diag_apply.argtypes = [C.c_void_p, f64p, f64p]
        L.oracle_ch_dF_dc.argtypes = [C.c_int64, f64p, f64p]
        L.oracle_ch_apply.argtypes = [C.c_void_p, f64p, f64p]
    return _libs[variant]


def _p(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(f64p)


def _pi(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(i64p)


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


# --- BLAS-1 ---------------------------------------------------------------------------
def dot(a, b) -> float:
    a, b = f64(a).ravel(), f64(b).ravel()
    return lib().oracle_dot(a.size, _p(a), _p(b))


def norm2(a) -> float:
    a = f64(a).ravel()
    return lib().oracle_norm2(a.size, _p(a))


def vsum(a) -> float:
    a = f64(a).ravel()
    return lib().oracle_sum(a.size, _p(a))


def norm1(a) -> float:
    a = f64(a).ravel()
    return lib().oracle_norm1(a.size, _p(a))


def norm_inf(a) -> float:
    a = f64(a).ravel()
    return lib().oracle_norm_inf(a.size, _p(a))


def rng_reset(variant: str = "strict") -> None:
    """Restart the oracle's function-static mt19937_64 (a fresh process in the reference's terms)."""
    lib(variant).oracle_rng_reset()


def fill_randomly(n: int) -> np.ndarray:
    out = np.empty(n)
    lib().oracle_fill_randomly(n, _p(out))
    return out


def safe_divide(x: float, y: float) -> float:
    return lib().oracle_safe_divide(x, y)


def sym_ortho(a: float, b: float):
    cs, sn, rr = C.c_double(), C.c_double(), C.c_double()
    lib().oracle_sym_ortho(a, b, C.byref(cs), C.byref(sn), C.byref(rr))
    return cs.value, sn.value, rr.value


def axpy(y, a, x):
    lib().oracle_axpy(y.size, _p(y), a, _p(x))


def axmy(y, a, x):
    lib().oracle_axmy(y.size, _p(y), a, _p(x))


def xpay(y, x, b):
    lib().oracle_xpay(y.size, _p(y), _p(x), b)


def bicg_p(p, r, b, w, v):
    lib().oracle_bicg_p(p.size, _p(p), _p(r), b, w, _p(v))


def sub_from(r, b):
    lib().oracle_sub_from(r.size, _p(r), _p(b))


def div_scalar(y, s):
    lib().oracle_div_scalar(y.size, _p(y), s)


def mul_scalar(y, s):
    lib().oracle_mul_scalar(y.size, _p(y), s)


def expr1(a, s, b, c) -> np.ndarray:
    a, b, c = f64(a).ravel(), f64(b).ravel(), f64(c).ravel()
    out = np.empty_like(a)
    lib().oracle_expr1(a.size, _p(out), _p(a), s, _p(b), _p(c))
    return out


def vdiv(s, a, b) -> np.ndarray:
    """``(s * a) / b`` elementwise, ``a is None``: ``s / b`` (MatrixMath.hpp:261-265, :298-302)."""
    b = f64(b).ravel()
    out = np.empty_like(b)
    lib().oracle_vdiv(b.size, _p(out), float(s), None if a is None else _p(f64(a).ravel()), _p(b))
    return out


# --- operators ------------------------------------------------------------------------
class Mesh:
    """Keeps the numpy arrays alive behind an ``oracle_mesh`` struct."""

    def __init__(self, g, ghost: Optional[np.ndarray] = None):
        self.g = g
        self._keep = dict(
            inner=np.ascontiguousarray(g.inner, np.int64), outer=np.ascontiguousarray(g.outer, np.int64),
            area=f64(g.area), center=f64(g.center), volume=f64(g.volume),
            b_cell=np.ascontiguousarray(g.b_cell, np.int64), b_area=f64(g.b_area),
            b_center=f64(g.b_center))
        k = self._keep
        self.c = _Mesh(g.n_total, g.n_faces, g.n_bfaces, g.dim, _pi(k["inner"]), _pi(k["outer"]),
                       _p(k["area"]), _p(k["center"]), _p(k["volume"]), _pi(k["b_cell"]),
                       _p(k["b_area"]), _p(k["b_center"]), None)
        if ghost is not None:
            k["ghost"] = f64(ghost)
            self.c.b_ghost = _p(k["ghost"])


class StencilOperator:
    """``y = beta*x + alpha*L(x) [+ conv*C_v(x)]`` through the reference-order face loops."""

    def __init__(self, g, alpha: float, beta: float, conv: float = 0.0, vel=(0.0, 0.0, 0.0),
                 variant: str = "strict"):
        self.variant = variant
        self.mesh = Mesh(g)
        self.n = g.n_total
        v = (C.c_double * 3)(*[float(t) for t in vel])
        self.c = _StencilOp(C.pointer(self.mesh.c), alpha, beta, conv, v)
        self.fn = C.cast(lib(variant).oracle_stencil_apply, C.c_void_p)
        self.ctx = C.cast(C.pointer(self.c), C.c_void_p)

    def apply(self, x: np.ndarray) -> np.ndarray:
        x = f64(x)
        y = np.empty_like(x)
        lib(self.variant).oracle_stencil_apply(self.ctx, _p(y), _p(x))
        return y


class CsrOperator:
    def __init__(self, a):
        a = a.tocsr()
        self.n = a.shape[0]
        self._keep = (np.ascontiguousarray(a.indptr, np.int64), np.ascontiguousarray(a.indices, np.int64),
                      f64(a.data))
        self.c = _CsrOp(self.n, _pi(self._keep[0]), _pi(self._keep[1]), _p(self._keep[2]))
        self.fn = C.cast(lib().oracle_csr_apply, C.c_void_p)
        self.ctx = C.cast(C.pointer(self.c), C.c_void_p)

    def apply(self, x):
        x = f64(x)
        y = np.empty_like(x)
        lib().oracle_csr_apply(self.ctx, _p(y), _p(x))
        return y


class _GatherOp(C.Structure):
    _fields_ = [("n", C.c_int64), ("width", C.c_int32), ("col", i64p), ("w", f64p), ("ext", f64p),
                ("alpha", C.c_double), ("beta", C.c_double), ("seed", C.c_uint64), ("applies", C.c_uint64)]


class GatherOperator:
    """INVESTIGATION only (tools/bicgstab_draw_study.py): the face-graph operator in the HIP kernels' arithmetic
    form -- pre-divided weights, rows gathered in face order, difference form, FMAs -- with an optional seeded
    perturbation of every apply by at most one unit in the last place (``oracle_gather_apply``)."""

    def __init__(self, g, alpha: float, beta: float, seed: int = 0, variant: str = "strict"):
        from stormruler_amd import mesh as _mesh

        coef, b_coef = _mesh.face_coefficients(g)
        n = g.n_cells
        inner, outer = np.asarray(g.inner, np.int64), np.asarray(g.outer, np.int64)
        wi, wo = coef / g.volume[inner], coef / g.volume[outer]  # storm_hip_op_create_from_faces
        # rows in face order: entry 2f = (inner -> outer), 2f + 1 = (outer -> inner); a stable sort by row keeps it
        rows = np.empty(2 * g.n_faces, np.int64)
        rows[0::2], rows[1::2] = inner, outer
        cols = np.empty_like(rows)
        cols[0::2], cols[1::2] = outer, inner
        vals = np.empty(2 * g.n_faces)
        vals[0::2], vals[1::2] = wi, wo
        order = np.argsort(rows, kind="stable")
        rows, cols, vals = rows[order], cols[order], vals[order]
        cnt = np.bincount(rows, minlength=n)
        width = int(cnt.max()) if n else 0
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        slot = np.arange(rows.size) - start[rows]
        self.col = np.full((n, width), -1, np.int64)
        self.w = np.zeros((n, width))
        self.col[rows, slot], self.w[rows, slot] = cols, vals
        ext = np.zeros(n)
        if g.n_bfaces:
            np.subtract.at(ext, np.asarray(g.b_cell, np.int64), b_coef / g.volume[np.asarray(g.b_cell, np.int64)])
        self.ext = ext
        self.n, self.variant = n, variant
        self.c = _GatherOp(n, width, _pi(self.col.reshape(-1)), _p(self.w.reshape(-1)), _p(self.ext), alpha, beta, seed, 0)
        self.fn = C.cast(lib(variant).oracle_gather_apply, C.c_void_p)
        self.ctx = C.cast(C.pointer(self.c), C.c_void_p)

    def apply(self, x: np.ndarray) -> np.ndarray:
        x = f64(x)
        y = np.empty_like(x)
        lib(self.variant).oracle_gather_apply(self.ctx, _p(y), _p(x))
        return y


class CallbackOperator:
    """Operator whose ``mul`` is a Python callable (used by the 2-rank gloo tests)."""

    def __init__(self, n: int, fn):
        self.n = n

        def _cb(_ctx, yp, xp):
            x = np.ctypeslib.as_array(xp, shape=(n,))
            y = np.ctypeslib.as_array(yp, shape=(n,))
            y[:] = fn(x)

        self._cb = APPLY_FN(_cb)
        self.fn = C.cast(self._cb, C.c_void_p)
        self.ctx = None


class _DiagOp(C.Structure):
    _fields_ = [("n", C.c_int64), ("d", f64p)]


class DiagOperator:
    """``y = d .* x`` (a Jacobi preconditioner passes ``d = 1 / diag(A)``)."""

    def __init__(self, d, variant: str = "strict"):
        self.d = f64(d)
        self.n = self.d.size
        self.c = _DiagOp(self.n, _p(self.d))
        self.fn = C.cast(lib(variant).oracle_diag_apply, C.c_void_p)
        self.ctx = C.cast(C.pointer(self.c), C.c_void_p)


@dataclass
class SolveResult:
    x: np.ndarray
    iterations: int
    absolute_error: float
    relative_error: float
    initial_error: float
    converged: bool
    num_applies: int
    history: np.ndarray


def solve(kind: str, op, b, x0=None, num_iterations: int = 2000, abs_tol: float = 1e-6,
          rel_tol: float = 1e-6, num_inner_iterations: int = 50, relaxation_factor: float = 1e-4,
          variant: str = "strict", pre=None, side: str = "right") -> SolveResult:
    """``solve<XSolver>(x, b, op)`` of Solvers/Solver.hpp:261-265 with the reference defaults; ``pre`` /
    ``side`` are the solver's ``pre_op`` / ``pre_side`` members (:74-75).  The number of preconditioner
    applications of the last solve is ``last_pre_applies()``."""
    b = f64(b)
    x = np.zeros_like(b) if x0 is None else f64(x0).copy()
    p = _Params(num_iterations, abs_tol, rel_tol, num_inner_iterations, relaxation_factor)
    r = _Result()
    hist = np.full(num_iterations + 1, np.nan)
    L = lib(variant)
    fn = {"cg": L.oracle_solve_cg, "bicgstab": L.oracle_solve_bicgstab, "gmres": L.oracle_solve_gmres,
          "richardson": L.oracle_solve_richardson, "cgs": L.oracle_solve_cgs, "tfqmr": L.oracle_solve_tfqmr,
          "tfqmr1": L.oracle_solve_tfqmr1, "bicgstabl": L.oracle_solve_bicgstabl, "idrs": L.oracle_solve_idrs}[kind]
    if pre is not None:
        L.oracle_set_preconditioner(pre.fn, pre.ctx, {"left": 0, "right": 1, "symmetric": 2}[side])
    fn(op.fn, op.ctx, b.size, _p(x), _p(b), C.byref(p), C.byref(r), _p(hist))
    return SolveResult(x, r.iterations, r.absolute_error, r.relative_error, r.initial_error,
                       bool(r.converged), r.num_applies, hist[: r.iterations + 1].copy())


def last_pre_applies(variant: str = "strict") -> int:
    return lib(variant).oracle_last_pre_applies()


SIDES = {"left": 0, "right": 1, "symmetric": 2}


def solve_gmres_pre(op, pre, b, x0=None, side: str = "right", flexible: bool = False, num_iterations: int = 2000,
                    abs_tol: float = 1e-6, rel_tol: float = 1e-6, num_inner_iterations: int = 50,
                    variant: str = "strict"):
    """Preconditioned GMRES / FGMRES (SolverGmres.hpp, the ``pre_op != nullptr`` branches).  ``pre`` is any
    operator object (``DiagOperator``, ``CallbackOperator`` ...) or ``None``.  Returns
    ``(SolveResult, n_preconditioner_applies)``."""
    b = f64(b)
    x = np.zeros_like(b) if x0 is None else f64(x0).copy()
    p = _Params(num_iterations, abs_tol, rel_tol, num_inner_iterations, 1e-4)
    r = _Result()
    hist = np.full(num_iterations + 1, np.nan)
    n_pre = lib(variant).oracle_solve_gmres_pre(op.fn, op.ctx, pre.fn if pre is not None else None,
                                                pre.ctx if pre is not None else None, SIDES[side], int(flexible),
                                                b.size, _p(x), _p(b), C.byref(p), C.byref(r), _p(hist))
    return SolveResult(x, r.iterations, r.absolute_error, r.relative_error, r.initial_error,
                       bool(r.converged), r.num_applies, hist[: r.iterations + 1].copy()), n_pre


def solve_jfnk(op, b, x0=None, num_iterations: int = 2000, abs_tol: float = 1e-6, rel_tol: float = 1e-6,
               variant: str = "strict"):
    """``solve<JfnkSolver>`` (SolverNewton.hpp:101-173); ``op`` may be nonlinear.  Returns
    ``(SolveResult, total inner BiCGStab iterations)``."""
    b = f64(b)
    x = np.zeros_like(b) if x0 is None else f64(x0).copy()
    p = _Params(num_iterations, abs_tol, rel_tol, 50, 1e-4)
    r = _Result()
    hist = np.full(num_iterations + 1, np.nan)
    inner = lib(variant).oracle_solve_jfnk(op.fn, op.ctx, b.size, _p(x), _p(b), C.byref(p), C.byref(r), _p(hist))
    return SolveResult(x, r.iterations, r.absolute_error, r.relative_error, r.initial_error,
                       bool(r.converged), r.num_applies, hist[: r.iterations + 1].copy()), inner


class _ChOp(C.Structure):
    _fields_ = [("mesh", C.POINTER(_Mesh)), ("f", f64p), ("c", f64p), ("w_hat", f64p),
                ("tau", C.c_double), ("Gamma", C.c_double), ("sigma", C.c_double)]


def ch_operator_apply(g_or_mesh, f, c, c_in, tau: float = 1.0e-3, Gamma: float = 1.0e-4, sigma: float = 2.0):
    """One application of the playground's operator lambda (Playground.cpp:153-167, constants of :113):
    returns ``(c_hat, w_hat)``."""
    m = g_or_mesh if isinstance(g_or_mesh, Mesh) else Mesh(g_or_mesh)
    f, c, c_in = f64(f), f64(c), f64(c_in)
    c_hat, w_hat = np.empty_like(c), np.empty_like(c)
    op = _ChOp(C.pointer(m.c), _p(f), _p(c), _p(w_hat), tau, Gamma, sigma)
    lib().oracle_ch_apply(C.byref(op), _p(c_hat), _p(c_in))
    return c_hat, w_hat


class ChOperator:
    """The playground's operator lambda (Playground.cpp:153-167) as an operator ``solve`` can take: what
    ``cahn_hilliard_step`` hands to ``solve<CgSolver>`` (:151).  ``f`` and ``c`` are captured by reference like the
    lambda's: the arrays passed in are the ones read at every application."""

    def __init__(self, g_or_mesh, f, c, tau: float = 1.0e-3, Gamma: float = 1.0e-4, sigma: float = 2.0, variant: str = "strict"):
        self.mesh = g_or_mesh if isinstance(g_or_mesh, Mesh) else Mesh(g_or_mesh)
        self.f, self.c = f64(f), f64(c)
        self.w_hat = np.empty_like(self.c)
        self.op = _ChOp(C.pointer(self.mesh.c), _p(self.f), _p(self.c), _p(self.w_hat), tau, Gamma, sigma)
        self.variant = variant
        self.fn = C.cast(lib(variant).oracle_ch_apply, C.c_void_p)
        self.ctx = C.cast(C.pointer(self.op), C.c_void_p)

    def apply(self, c_in):
        c_in = f64(c_in)
        c_hat = np.empty_like(c_in)
        lib(self.variant).oracle_ch_apply(self.ctx, _p(c_hat), _p(c_in))
        return c_hat


def cahn_hilliard_step(g_or_mesh, c, **solver_knobs):
    """``cahn_hilliard_step`` (Playground.cpp:133-174): ``f <<= map(dF_dc, c)``; ``c_hat <<= c`` (the warm start, :150);
    ``solve<CgSolver>(c_hat, c, lambda)``.  Returns ``(c_hat, SolveResult)``."""
    c = f64(c)
    op = ChOperator(g_or_mesh, dF_dc(c), c)
    res = solve("cg", op, c, x0=c, **solver_knobs)
    return res.x, res


def solve_non_uniform(kind: str, op, b, x0=None, **knobs) -> SolveResult:
    """``solve_non_uniform(solver, x, b, op)`` (Solvers/Solver.hpp:271-292) for an operator with ``op.apply``: ``z = A(0)``;
    ``f = b - z``; solve with the operator ``y = A(x); y -= z``."""
    b = f64(b)
    z = op.apply(np.zeros_like(b))                               # :279-282
    f = b - z                                                    # :283
    uni = CallbackOperator(b.size, lambda x: op.apply(x) - z)    # :285-289
    return solve(kind, uni, f, x0=x0, **knobs)                   # :291


def cahn_hilliard_step_non_uniform(g_or_mesh, c, **solver_knobs):
    """``cahn_hilliard_step`` (Playground.cpp:133-174) with ``solve_non_uniform`` in place of ``solve``: what the affine
    lambda calls for.  Returns ``(c_hat, SolveResult)``."""
    c = f64(c)
    res = solve_non_uniform("cg", ChOperator(g_or_mesh, dF_dc(c), c), c, x0=c, **solver_knobs)
    return res.x, res


def dF_dc(c):
    """``map(dF_dc, c)``, Playground.cpp:142-148."""
    c = f64(c)
    f = np.empty_like(c)
    lib().oracle_ch_dF_dc(c.size, _p(f), _p(c))
    return f


_omp_lib = None


def _omp():
    global _omp_lib
    if _omp_lib is None:
        build()
        _omp_lib = C.CDLL(os.path.join(_HERE, "liboracle_omp.so"))
        _omp_lib.oracle_omp_cg.restype = C.c_double
        _omp_lib.oracle_omp_cg.argtypes = [C.c_int64, i64p, C.POINTER(C.c_int32), f64p, f64p, f64p, C.c_int64, C.c_int]
        _omp_lib.oracle_omp_cg_box.restype = C.c_double
        _omp_lib.oracle_omp_cg_box.argtypes = [C.c_int, C.c_int64, C.c_int, f64p]
    return _omp_lib


def omp_cg_box(n: int, iterations: int, threads: int = 0):
    """``iterations`` OpenMP CG steps on the n^3 Dirichlet box (b = 1, x0 = 0), matrix built inside with
    first-touch placement.  Returns ``(|r| after the last step, seconds of the iteration loop, threads)``."""
    L = _omp()
    sec = C.c_double(0.0)
    res = L.oracle_omp_cg_box(int(n), int(iterations), int(threads), C.byref(sec))
    return res, sec.value, (threads if threads > 0 else L.oracle_omp_max_threads())


def omp_cg(a_csr, b, iterations: int, threads: int = 0):
    """OpenMP-parallel CG sample on assembled CSR rows (oracle/storm_oracle_omp.c: NOT the reference's loop
    order; the optional "what the host CPU could do" line of bench.py).  Returns ``(x, |r|, threads used)``."""
    _omp()
    a = a_csr.tocsr()
    rp = np.ascontiguousarray(a.indptr, np.int64)
    col = np.ascontiguousarray(a.indices, np.int32)
    val = f64(a.data)
    b = f64(b)
    x = np.zeros_like(b)
    res = _omp_lib.oracle_omp_cg(a.shape[0], _pi(rp), col.ctypes.data_as(C.POINTER(C.c_int32)), _p(val), _p(x), _p(b),
                                 int(iterations), int(threads))
    return x, res, (threads if threads > 0 else _omp_lib.oracle_omp_max_threads())
