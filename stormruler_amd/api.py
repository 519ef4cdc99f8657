"""Host-side mirror of the reference's Operator / Vector / Solver interface over the C ABI.

Same names, argument meaning and error behaviour as the reference so tests read like code
written against ``Storm::``:

* ``DeviceVector``           <- ``Feathers::Field`` as the solver ``Vector`` (Feathers/Field.hpp:60-114)
* ``Operator`` / ``FunctionalOperator`` / ``make_operator`` / ``make_symmetric_operator``
                             <- Solvers/Operator.hpp:66-200
* ``IterativeSolver`` / ``InnerOuterIterativeSolver`` / ``CgSolver`` / ``BiCgStabSolver`` / ``GmresSolver``
  / ``solve``                <- Solvers/Solver.hpp:43-292, SolverCg.hpp, SolverBiCgStab.hpp, SolverGmres.hpp
* ``dot_product`` / ``norm_2`` / ``fill_with``   <- Bittern/MatrixAlgorithms.hpp:262-317

Every vector statement a solver body executes (``x += alpha * p``, ``p <<= r + beta * p`` ...)
lowers to exactly one C-ABI call; an expression form without a kernel raises instead of
falling back to a host loop.  With a :class:`HipStencilOperator` and no preconditioner the
solver classes hand the whole solve to the device-resident entry points
(``storm_hip_solve_*``); with any other ``Operator`` (e.g. a Python lambda through
``make_operator``) they run the reference's statement sequence over the BLAS-1 calls.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
import weakref
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import check, lib
from .mesh import FaceGraph, face_coefficients

_LOG = logging.getLogger("stormruler_amd.solvers")

real_t = float

# ---------------------------------------------------------------------------------------------


class Context:
    """One GPU, its streams and reduction workspace (one per process / rank)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        check(lib.storm_hip_ctx_create(device, C.byref(h)))
        self._h = h
        self.n_ranks, self.rank = 1, 0
        self._children = weakref.WeakSet()  # vectors / operators living on this context

    def close(self):
        """Destroy the context after every vector and operator created on it."""
        if getattr(self, "_h", None):
            for child in list(self._children):
                child._free()
            lib.storm_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib.storm_hip_ctx_sync(self._h))

    def info(self):
        name = C.create_string_buffer(128)
        cus, mem = C.c_int(), C.c_int64()
        check(lib.storm_hip_ctx_info(self._h, name, 128, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "num_cus": cus.value, "total_mem": mem.value}

    def set_option(self, key: str, value: int):
        check(lib.storm_hip_ctx_set_option(self._h, key.encode(), int(value)))

    def spmv_profile(self):
        """(launches, total_ms, min_ms) of the SpMV kernel since the last call (option profile_spmv)."""
        n, tot, mn = C.c_int64(), C.c_double(), C.c_double()
        check(lib.storm_hip_ctx_get_spmv_profile(self._h, C.byref(n), C.byref(tot), C.byref(mn)))
        return n.value, tot.value, mn.value

    def timer_start(self):
        check(lib.storm_hip_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(lib.storm_hip_timer_stop(self._h, C.byref(ms)))
        return ms.value

    # -- multi-GPU ---------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(lib.storm_hip_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: Optional[bytes], n_ranks: int, rank: int):
        buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        check(lib.storm_hip_ctx_comm_init(self._h, buf, n_ranks, rank))
        self.n_ranks, self.rank = n_ranks, rank

    def comm_init_host(self, n_ranks: int, rank: int, allreduce, exchange):
        """Host-staged transport (``storm_hip_ctx_comm_init_host``): ``allreduce(buf)`` sums a float64 array over
        the ranks in place; ``exchange(nbr_rank, sends, recvs)`` delivers ``sends[q]`` to rank ``nbr_rank[q]``
        and fills ``recvs[q]`` with what that rank sent here (lists of float64 arrays)."""

        def _allreduce(_user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:  # never unwind through the C frames
                import traceback

                traceback.print_exc()
                return 1

        def _exchange(_user, n_nbrs, nbr_rank, send_ptr, send, recv_ptr, recv):
            try:
                sp = np.ctypeslib.as_array(send_ptr, shape=(n_nbrs + 1,))
                rp = np.ctypeslib.as_array(recv_ptr, shape=(n_nbrs + 1,))
                nb = np.ctypeslib.as_array(nbr_rank, shape=(n_nbrs,))
                sbuf = np.ctypeslib.as_array(send, shape=(max(int(sp[-1]), 1),))
                rbuf = np.ctypeslib.as_array(recv, shape=(max(int(rp[-1]), 1),))
                exchange([int(r) for r in nb], [sbuf[sp[q]:sp[q + 1]] for q in range(n_nbrs)],
                         [rbuf[rp[q]:rp[q + 1]] for q in range(n_nbrs)])
                return 0
            except Exception:
                import traceback

                traceback.print_exc()
                return 1

        self._host_cbs = (_lib.ALLREDUCE_FN(_allreduce), _lib.EXCHANGE_FN(_exchange))  # keep alive
        check(lib.storm_hip_ctx_comm_init_host(self._h, n_ranks, rank, self._host_cbs[0], self._host_cbs[1], None))
        self.n_ranks, self.rank = n_ranks, rank


# ---------------------------------------------------------------------------------------------
# Expression nodes (the build's counterparts of Bittern's lazy MapMatrixView nodes,
# Bittern/MatrixMath.hpp:44-105,247-285): just enough structure for one kernel per statement.


class _Scaled:  # a * v
    def __init__(self, a: float, v: "DeviceVector"):
        self.a, self.v = float(a), v

    def __add__(self, o):  # omega * v + gamma * u   (SolverIdrs.hpp:208)
        if isinstance(o, _Scaled):
            return _Lin2(self.a, self.v, o.a, o.v)
        return NotImplemented


class _Quot:  # v / s  (true elementwise division, SolverIdrs.hpp:131)
    def __init__(self, v: "DeviceVector", s: float):
        self.v, self.s = v, float(s)


class _Lin2:  # a*x + b*z
    def __init__(self, a, x, b, z):
        self.a, self.x, self.b, self.z = float(a), x, float(b), z

    def __rmul__(self, s):
        return _ScaledLin2(float(s), self)


class _ScaledLin2:  # s * (a*x + b*z)
    def __init__(self, s, lin):
        self.s, self.lin = s, lin


class _Lin3:  # r + s * (a*x + b*z)
    def __init__(self, r, s, lin):
        self.r, self.s, self.lin = r, s, lin


class DeviceVector:
    """N doubles in HBM (+ halo tail); the solver ``Vector`` (concept legacy_vector_like,
    Solvers/Operator.hpp:39-45)."""

    __array_ufunc__ = None  # numpy scalars defer to __rmul__ instead of broadcasting over the object

    def __init__(self, ctx: Optional[Context] = None, n_owned: int = 0, n_halo: int = 0):
        self.ctx = ctx
        self._h = None
        if ctx is not None:
            h = C.c_void_p()
            check(lib.storm_hip_vec_create(ctx._h, n_owned, n_halo, C.byref(h)))
            self._h = h
            ctx._children.add(self)
        self.n_owned, self.n_halo = n_owned, n_halo

    # Field::assign(other, copy): allocate like `other`, zero-initialised; `copy` is ignored by the
    # reference (Feathers/Field.hpp:82-84) and therefore here.
    def assign(self, other: "DeviceVector", copy: bool = True) -> None:
        self._free()
        h = C.c_void_p()
        check(lib.storm_hip_vec_create_like(other._h, C.byref(h)))
        self._h, self.ctx = h, other.ctx
        self.ctx._children.add(self)
        self.n_owned, self.n_halo = other.n_owned, other.n_halo

    def shape(self):
        return (self.n_owned, 1)  # Field::shape() = {N, NumVars}, Field.hpp:77-79

    def _free(self):
        if getattr(self, "_h", None):
            lib.storm_hip_vec_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self._free()
        except Exception:
            pass

    @classmethod
    def from_numpy(cls, ctx: Context, a: np.ndarray, n_halo: int = 0) -> "DeviceVector":
        a = np.ascontiguousarray(a, np.float64)
        v = cls(ctx, a.size, n_halo)
        check(lib.storm_hip_vec_upload(v._h, a.ctypes.data_as(_lib.f64p), a.size))
        return v

    def upload(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a, np.float64)
        check(lib.storm_hip_vec_upload(self._h, a.ctypes.data_as(_lib.f64p), a.size))

    def to_numpy(self, with_halo: bool = False) -> np.ndarray:
        n = self.n_owned + (self.n_halo if with_halo else 0)
        out = np.empty(n, np.float64)
        check(lib.storm_hip_vec_download(self._h, out.ctypes.data_as(_lib.f64p), n))
        return out

    # -- expression builders ---------------------------------------------------------------
    def __rmul__(self, a):  # alpha * v          MatrixMath.hpp:247-256
        return _Scaled(a, self)

    def __add__(self, o):  # MatrixMath.hpp:273-279
        if isinstance(o, DeviceVector):
            return _Lin2(1.0, self, 1.0, o)
        if isinstance(o, _Scaled):
            return _Lin2(1.0, self, o.a, o.v)
        if isinstance(o, _ScaledLin2):
            return _Lin3(self, o.s, o.lin)
        return NotImplemented

    def __truediv__(self, s):
        return _Quot(self, s)

    def __sub__(self, o):  # MatrixMath.hpp:280-285
        if isinstance(o, DeviceVector):
            return _Lin2(1.0, self, -1.0, o)
        if isinstance(o, _Scaled):
            return _Lin2(1.0, self, -o.a, o.v)
        return NotImplemented

    # -- targets (TargetMatrixInterface, Bittern/MatrixTarget.hpp:53-136) -------------------
    def __ilshift__(self, e):  # out <<= expr     MatrixAlgorithms.hpp:120-124
        if isinstance(e, DeviceVector):
            check(lib.storm_hip_copy(self._h, e._h))
        elif isinstance(e, _Scaled):
            check(lib.storm_hip_axpbz(self._h, e.a, e.v._h, 0.0, e.v._h))
        elif isinstance(e, _Lin2):
            check(lib.storm_hip_axpbz(self._h, e.a, e.x._h, e.b, e.z._h))
        elif isinstance(e, _Quot):
            if e.v is not self:
                check(lib.storm_hip_copy(self._h, e.v._h))
            check(lib.storm_hip_div_scalar(self._h, e.s))
        elif isinstance(e, _Lin3):
            # p <<= r + beta * (p - omega * v)  (SolverBiCgStab.hpp:119),  p <<= u + beta * (q + beta * p)
            # (SolverCgs.hpp:122): one kernel evaluating r + s * (a x + b z) in that nesting
            check(lib.storm_hip_lin3(self._h, e.r._h, e.s, e.lin.a, e.lin.x._h, e.lin.b, e.lin.z._h))
        elif isinstance(e, _ScaledLin2):
            # z <<= delta_inverse * (z - w)  (SolverNewton.hpp:148): the inner sum is rounded before
            # the scaling, as the reference's expression tree evaluates it
            check(lib.storm_hip_axpbz(self._h, e.lin.a, e.lin.x._h, e.lin.b, e.lin.z._h))
            check(lib.storm_hip_scale(self._h, e.s))
        else:
            raise NotImplementedError(f"no device kernel for `<<=` of {type(e).__name__}")
        return self

    def __iadd__(self, e):  # MatrixTarget.hpp:108-113
        if isinstance(e, _Scaled):
            check(lib.storm_hip_axpy(self._h, e.a, e.v._h))
        elif isinstance(e, DeviceVector):
            check(lib.storm_hip_axpy(self._h, 1.0, e._h))
        else:
            raise NotImplementedError(f"no device kernel for `+=` of {type(e).__name__}")
        return self

    def __isub__(self, e):  # MatrixTarget.hpp:114-119
        if isinstance(e, _Scaled):
            check(lib.storm_hip_axpy(self._h, -e.a, e.v._h))
        elif isinstance(e, DeviceVector):
            check(lib.storm_hip_axpy(self._h, -1.0, e._h))
        else:
            raise NotImplementedError(f"no device kernel for `-=` of {type(e).__name__}")
        return self

    def __imul__(self, s):  # MatrixTarget.hpp:96-99
        check(lib.storm_hip_scale(self._h, float(s)))
        return self

    def __itruediv__(self, s):  # MatrixTarget.hpp:101-105
        check(lib.storm_hip_div_scalar(self._h, float(s)))
        return self


def dot_product(a: DeviceVector, b: DeviceVector) -> float:
    """Bittern/MatrixAlgorithms.hpp:310-317 (summed over all ranks)."""
    out = C.c_double()
    check(lib.storm_hip_dot(a._h, b._h, C.byref(out)))
    return out.value


def norm_2(a: DeviceVector) -> float:
    """Bittern/MatrixAlgorithms.hpp:262-270."""
    out = C.c_double()
    check(lib.storm_hip_norm2(a._h, C.byref(out)))
    return out.value


def fill_with(a: DeviceVector, value: float) -> None:
    """ADL hook the solver bodies call (Solvers/Solver.hpp:281, SolverBiCgStab.hpp:224)."""
    check(lib.storm_hip_fill(a._h, float(value)))


def vmul_add(y: DeviceVector, s: float, a: DeviceVector, b: DeviceVector) -> None:
    """``y += s * (a .* b)`` elementwise (nonlinear terms of a time-step driver)."""
    check(lib.storm_hip_vmul_add(y._h, float(s), a._h, b._h))


def vmul(y: DeviceVector, a: DeviceVector, b: DeviceVector) -> None:
    """``y = a .* b`` elementwise (a diagonal preconditioner's ``mul``)."""
    check(lib.storm_hip_vmul(y._h, a._h, b._h))


def fill_randomly(a: DeviceVector) -> None:
    """Bittern/MatrixAlgorithms.hpp:140-153 (same engine / distribution / sequence as the reference)."""
    check(lib.storm_hip_fill_randomly(a._h))


def rng_reset() -> None:
    lib.storm_hip_rng_reset()


def multi_dot(a: DeviceVector, bs: Sequence[DeviceVector]) -> np.ndarray:
    k = len(bs)
    arr = (C.c_void_p * k)(*[b._h for b in bs])
    out = np.empty(k)
    check(lib.storm_hip_multi_dot(a._h, arr, k, out.ctypes.data_as(_lib.f64p)))
    return out


def multi_axpy(y: DeviceVector, coefs: Sequence[float], xs: Sequence[DeviceVector]) -> None:
    k = len(xs)
    arr = (C.c_void_p * k)(*[x._h for x in xs])
    cf = np.ascontiguousarray(coefs, np.float64)
    check(lib.storm_hip_multi_axpy(y._h, cf.ctypes.data_as(_lib.f64p), arr, k))


def safe_divide(x: float, y: float) -> float:
    """Crow/MathUtils.hpp:49-52."""
    return 0.0 if y == 0.0 else x / y


def sym_ortho(a: float, b: float):
    """Crow/MathUtils.hpp:164-179."""
    rr = math.hypot(a, b)
    if rr > 0.0:
        return a / rr, b / rr, rr
    return 1.0, 0.0, rr


# ---------------------------------------------------------------------------------------------
# Operators (Solvers/Operator.hpp)


class Operator:
    """Abstract operator y <- A(x)  (Operator.hpp:66-120)."""

    def mul(self, y_vec: DeviceVector, x_vec: DeviceVector) -> None:  # :74
        raise NotImplementedError

    def mul_chain(self, z_vec, y_vec, other_op: "Operator", x_vec) -> None:  # :82-88
        other_op.mul(y_vec, x_vec)
        self.mul(z_vec, y_vec)

    def Residual(self, r_vec: DeviceVector, b_vec: DeviceVector, x_vec: DeviceVector) -> None:  # :95-99
        self.mul(r_vec, x_vec)
        r_vec <<= b_vec - r_vec

    def ResidualNorm(self, b_vec: DeviceVector, x_vec: DeviceVector) -> float:  # :105-110
        r_vec = DeviceVector()
        r_vec.assign(b_vec, False)
        self.Residual(r_vec, b_vec, x_vec)
        return norm_2(r_vec)

    def conj_mul(self, x_vec, y_vec) -> None:  # :116-118
        raise RuntimeError("`Operator::conj_mul` was not overriden")


class FunctionalOperator(Operator):
    """Operator.hpp:125-168."""

    def __init__(self, mat_vec_func: Callable, conj_mat_vec_func: Optional[Callable] = None):
        assert mat_vec_func is not None
        self._mat_vec, self._conj = mat_vec_func, conj_mat_vec_func

    def mul(self, y_vec, x_vec):
        self._mat_vec(y_vec, x_vec)

    def conj_mul(self, x_vec, y_vec):
        if self._conj is None:
            raise RuntimeError("`FunctionalOperator::conj_mul` conjugate product function was not set.")
        self._conj(x_vec, y_vec)


def make_operator(mat_vec_func, conj_mat_vec_func=None) -> FunctionalOperator:  # Operator.hpp:177-190
    return FunctionalOperator(mat_vec_func, conj_mat_vec_func)


def make_symmetric_operator(mat_vec_func) -> FunctionalOperator:  # Operator.hpp:196-200
    return FunctionalOperator(mat_vec_func, mat_vec_func)


class StencilMatrix:
    """The device-resident face-graph operator M (sliced ELL + CSR tail); owns the handle."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self._h = ctx, handle
        ctx._children.add(self)

    def _free(self):
        self.close()

    @classmethod
    def from_face_graph(cls, ctx: Context, g: FaceGraph) -> "StencilMatrix":
        """Diffusion stencil of ``stormDivGrad`` (Playground.cpp:115-131) from mesh quantities."""
        coef, b_coef = face_coefficients(g)
        h = C.c_void_p()
        i64, f64 = _lib.i64p, _lib.f64p
        inner = np.ascontiguousarray(g.inner, np.int64)
        outer = np.ascontiguousarray(g.outer, np.int64)
        b_cell = np.ascontiguousarray(g.b_cell, np.int64)
        vol = np.ascontiguousarray(g.volume, np.float64)
        coef = np.ascontiguousarray(coef)
        b_coef = np.ascontiguousarray(b_coef)
        check(lib.storm_hip_op_create_from_faces(
            ctx._h, g.n_cells, g.n_halo, g.n_faces, inner.ctypes.data_as(i64), outer.ctypes.data_as(i64),
            coef.ctypes.data_as(f64), g.n_bfaces, b_cell.ctypes.data_as(i64), b_coef.ctypes.data_as(f64),
            vol.ctypes.data_as(f64), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_face_weights(cls, ctx: Context, n_owned: int, n_halo: int, inner, outer, w_inner, w_outer,
                          diag_extra=None) -> "StencilMatrix":
        h = C.c_void_p()
        i64, f64 = _lib.i64p, _lib.f64p
        inner = np.ascontiguousarray(inner, np.int64)
        outer = np.ascontiguousarray(outer, np.int64)
        w_inner = np.ascontiguousarray(w_inner, np.float64)
        w_outer = np.ascontiguousarray(w_outer, np.float64)
        de = None if diag_extra is None else np.ascontiguousarray(diag_extra, np.float64)
        check(lib.storm_hip_op_create_from_face_weights(
            ctx._h, n_owned, n_halo, inner.size, inner.ctypes.data_as(i64), outer.ctypes.data_as(i64),
            w_inner.ctypes.data_as(f64), w_outer.ctypes.data_as(f64),
            None if de is None else de.ctypes.data_as(f64), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_csr(cls, ctx: Context, a, n_halo: int = 0) -> "StencilMatrix":
        a = a.tocsr()
        rp = np.ascontiguousarray(a.indptr, np.int64)
        col = np.ascontiguousarray(a.indices, np.int64)
        val = np.ascontiguousarray(a.data, np.float64)
        h = C.c_void_p()
        check(lib.storm_hip_op_create_csr(ctx._h, a.shape[0], n_halo, rp.ctypes.data_as(_lib.i64p),
                                          col.ctypes.data_as(_lib.i64p), val.ctypes.data_as(_lib.f64p), C.byref(h)))
        return cls(ctx, h)

    def set_halo(self, nbr_rank, send_ptr, send_idx, recv_ptr) -> None:
        nbr = np.ascontiguousarray(nbr_rank, np.int32)
        sp = np.ascontiguousarray(send_ptr, np.int64)
        si = np.ascontiguousarray(send_idx, np.int64)
        rp = np.ascontiguousarray(recv_ptr, np.int64)
        check(lib.storm_hip_op_set_halo(self._h, nbr.size, nbr.ctypes.data_as(_lib.i32p), sp.ctypes.data_as(_lib.i64p),
                                        si.ctypes.data_as(_lib.i64p), rp.ctypes.data_as(_lib.i64p)))

    def stats(self) -> dict:
        s = _lib.OpStats()
        check(lib.storm_hip_op_get_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in s._fields_}

    def apply(self, alpha: float, beta: float, x: DeviceVector, y: DeviceVector) -> None:
        check(lib.storm_hip_op_apply(self._h, alpha, beta, x._h, y._h))

    def apply_add(self, alpha: float, x: DeviceVector, y: DeviceVector) -> None:
        """``y += alpha * M(x)``: ``stormDivGrad(mesh, y, alpha, x)`` (Playground.cpp:115-131) in its own form."""
        check(lib.storm_hip_op_apply_add(self._h, float(alpha), x._h, y._h))

    def diagonal(self, alpha: float, beta: float, d: DeviceVector, invert: bool = False) -> None:
        """``d`` <- the diagonal of ``beta*I + alpha*M`` (its safe inverse with ``invert``)."""
        check(lib.storm_hip_op_get_diagonal(self._h, float(alpha), float(beta), int(invert), d._h))

    def close(self):
        if getattr(self, "_h", None):
            lib.storm_hip_op_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class HipStencilOperator(Operator):
    """``A = beta*I + alpha*M`` as an ``Operator<DeviceVector>``."""

    def __init__(self, matrix: StencilMatrix, alpha: float = -1.0, beta: float = 0.0):
        self.matrix, self.alpha, self.beta = matrix, float(alpha), float(beta)

    def mul(self, y_vec: DeviceVector, x_vec: DeviceVector) -> None:
        self.matrix.apply(self.alpha, self.beta, x_vec, y_vec)

    def conj_mul(self, x_vec, y_vec):
        raise RuntimeError("`Operator::conj_mul` was not overriden")


# ---------------------------------------------------------------------------------------------
# Solvers (Solvers/Solver.hpp)


class PreconditionerSide:  # Preconditioner.hpp:39-58
    Left, Right, Symmetric = range(3)


class Preconditioner(Operator):  # Preconditioner.hpp:63-78
    def build(self, x_vec, b_vec, any_op) -> None:
        pass

    def add_secant(self, y_vec, s_vec) -> None:  # :76
        pass


class IdentityPreconditioner(Preconditioner):  # Preconditioner.hpp:84-97
    def mul(self, y_vec, x_vec):
        y_vec <<= x_vec

    def conj_mul(self, x_vec, y_vec):
        x_vec <<= y_vec


def stormDivGrad(matrix: StencilMatrix, u: DeviceVector, dt: float, c: DeviceVector) -> None:
    """``stormDivGrad(mesh, u, dt, c)``, source_apps/playground/Playground.cpp:115-131: ``u += dt * div grad c``
    (``matrix`` holds what the face loop reads from ``mesh``)."""
    matrix.apply_add(dt, c, u)


class JacobiPreconditioner(Preconditioner):
    """Diagonal preconditioner ``P = diag(A)^-1`` for a :class:`HipStencilOperator`, entirely on the
    device.  The build's own addition behind the reference's ``pre_op`` hook (the reference ships only
    the identity, Preconditioner.hpp:84-97): ``build`` (:70-72) reads the diagonal of the operator it is
    given, ``mul`` is one elementwise product."""

    def __init__(self):
        self._dinv: Optional[DeviceVector] = None

    def build(self, x_vec, b_vec, any_op) -> None:
        if not isinstance(any_op, HipStencilOperator):
            raise TypeError("JacobiPreconditioner needs a HipStencilOperator to read the diagonal from")
        self._dinv = _like(x_vec)
        any_op.matrix.diagonal(any_op.alpha, any_op.beta, self._dinv, invert=True)

    def mul(self, y_vec, x_vec):
        vmul(y_vec, self._dinv, x_vec)

    def conj_mul(self, x_vec, y_vec):
        vmul(x_vec, self._dinv, y_vec)


class Solver:  # Solver.hpp:43-57
    def solve(self, x_vec: DeviceVector, b_vec: DeviceVector, any_op: Operator) -> bool:
        raise NotImplementedError


class IterativeSolver(Solver):
    """Solver.hpp:62-149 -- same public knobs, defaults and convergence rule."""

    _native = None  # name of the whole-solver C entry point, if any

    def __init__(self):
        self.iteration = 0
        self.num_iterations = 2000
        self.absolute_error = 0.0
        self.relative_error = 0.0
        self.absolute_error_tolerance = 1.0e-6
        self.relative_error_tolerance = 1.0e-6
        self.pre_side = PreconditionerSide.Right
        self.pre_op: Optional[Preconditioner] = None
        self.name = ""
        # extras of this build
        self.check_lag = 0
        self.record_history = False
        self.history: Optional[np.ndarray] = None
        self.num_applies = 0
        self.initial_error = 0.0

    def init(self, x_vec, b_vec, any_op, pre_op) -> float:
        raise NotImplementedError

    def iterate(self, x_vec, b_vec, any_op, pre_op) -> float:
        raise NotImplementedError

    def finalize(self, x_vec, b_vec, any_op, pre_op) -> None:
        pass

    def _params(self) -> _lib.SolverParams:
        p = _lib.SolverParams()
        lib.storm_hip_solver_params_default(C.byref(p))
        p.num_iterations = self.num_iterations
        p.absolute_error_tolerance = self.absolute_error_tolerance
        p.relative_error_tolerance = self.relative_error_tolerance
        p.check_lag = self.check_lag
        return p

    def _solve_native(self, x_vec, b_vec, op: HipStencilOperator) -> bool:
        p = self._params()
        r = _lib.SolverResult()
        hist = np.zeros(self.num_iterations + 1) if self.record_history else None
        fn = getattr(lib, self._native)
        check(fn(op.matrix._h, op.alpha, op.beta, b_vec._h, x_vec._h, C.byref(p), C.byref(r),
                 None if hist is None else hist.ctypes.data_as(_lib.f64p)))
        self.iteration = r.iterations
        self.absolute_error, self.relative_error = r.absolute_error, r.relative_error
        self.initial_error, self.num_applies = r.initial_error, r.num_applies
        self.history = None if hist is None else hist[: r.iterations + 1]
        self._log()
        return bool(r.converged)

    def _log(self) -> None:
        """The reference's one line per solve (`STORM_INFO`, Solver.hpp:144-145) on the logger
        ``stormruler_amd.solvers`` at INFO level."""
        _LOG.info("n_iter: %4d, abs_err: %-12e, rel_err: %-12e", self.iteration, self.absolute_error, self.relative_error)

    def solve(self, x_vec, b_vec, any_op) -> bool:  # Solver.hpp:116-147
        if self._native and isinstance(any_op, HipStencilOperator) and self.pre_op is None:
            return self._solve_native(x_vec, b_vec, any_op)
        if self.pre_op is not None:
            self.pre_op.build(x_vec, b_vec, any_op)
        initial_error = self.init(x_vec, b_vec, any_op, self.pre_op)
        self.initial_error = initial_error
        self.absolute_error = initial_error
        hist = [initial_error]
        if self.absolute_error_tolerance > 0.0 and self.absolute_error < self.absolute_error_tolerance:
            self.finalize(x_vec, b_vec, any_op, self.pre_op)
            self.history = np.array(hist)
            return True
        converged = False
        self.iteration = 0
        while (not converged) and self.iteration < self.num_iterations:
            self.absolute_error = self.iterate(x_vec, b_vec, any_op, self.pre_op)
            self.relative_error = self.absolute_error / initial_error
            hist.append(self.absolute_error)
            converged |= (self.absolute_error_tolerance > 0.0) and (self.absolute_error < self.absolute_error_tolerance)
            converged |= (self.relative_error_tolerance > 0.0) and (self.relative_error < self.relative_error_tolerance)
            self.iteration += 1
        self.finalize(x_vec, b_vec, any_op, self.pre_op)
        self.history = np.array(hist)
        self._log()
        return converged


class InnerOuterIterativeSolver(IterativeSolver):
    """Solver.hpp:154-259."""

    def __init__(self):
        super().__init__()
        self.inner_iteration = 0
        self.num_inner_iterations = 50

    def _params(self):
        p = super()._params()
        p.num_inner_iterations = self.num_inner_iterations
        return p

    def outer_init(self, x_vec, b_vec, any_op, pre_op) -> float:
        raise NotImplementedError

    def inner_init(self, x_vec, b_vec, any_op, pre_op) -> None:
        pass

    def inner_iterate(self, x_vec, b_vec, any_op, pre_op) -> float:
        raise NotImplementedError

    def inner_finalize(self, x_vec, b_vec, any_op, pre_op) -> None:
        pass

    def outer_finalize(self, x_vec, b_vec, any_op, pre_op) -> None:
        pass

    def init(self, x_vec, b_vec, any_op, pre_op):  # :230-234
        return self.outer_init(x_vec, b_vec, any_op, pre_op)

    def iterate(self, x_vec, b_vec, any_op, pre_op):  # :236-248
        self.inner_iteration = self.iteration % self.num_inner_iterations
        if self.inner_iteration == 0:
            self.inner_init(x_vec, b_vec, any_op, pre_op)
        residual_norm = self.inner_iterate(x_vec, b_vec, any_op, pre_op)
        if self.inner_iteration == self.num_inner_iterations - 1:
            self.inner_finalize(x_vec, b_vec, any_op, pre_op)
        return residual_norm

    def finalize(self, x_vec, b_vec, any_op, pre_op):  # :250-257
        if self.inner_iteration != self.num_inner_iterations - 1:
            self.inner_finalize(x_vec, b_vec, any_op, pre_op)
        self.outer_finalize(x_vec, b_vec, any_op, pre_op)


class CgSolver(IterativeSolver):
    """SolverCg.hpp:47-128."""

    _native = "storm_hip_solve_cg"

    def init(self, x_vec, b_vec, lin_op, pre_op):  # :54-84
        self._p_vec, self._r_vec, self._z_vec = DeviceVector(), DeviceVector(), DeviceVector()
        self._p_vec.assign(x_vec, False)
        self._r_vec.assign(x_vec, False)
        self._z_vec.assign(x_vec, False)
        lin_op.Residual(self._r_vec, b_vec, x_vec)
        if pre_op is not None:
            pre_op.mul(self._z_vec, self._r_vec)
            self._p_vec <<= self._z_vec
            self._gamma = dot_product(self._r_vec, self._z_vec)
        else:
            self._p_vec <<= self._r_vec
            self._gamma = dot_product(self._r_vec, self._r_vec)
        return norm_2(self._r_vec) if pre_op is not None else math.sqrt(self._gamma)

    def iterate(self, x_vec, b_vec, lin_op, pre_op):  # :86-126
        lin_op.mul(self._z_vec, self._p_vec)
        alpha = safe_divide(self._gamma, dot_product(self._p_vec, self._z_vec))
        x_vec += alpha * self._p_vec
        self._r_vec -= alpha * self._z_vec
        gamma_bar = self._gamma
        if pre_op is not None:
            pre_op.mul(self._z_vec, self._r_vec)
            self._gamma = dot_product(self._r_vec, self._z_vec)
        else:
            self._gamma = dot_product(self._r_vec, self._r_vec)
        beta = safe_divide(self._gamma, gamma_bar)
        self._p_vec <<= (self._z_vec if pre_op is not None else self._r_vec) + beta * self._p_vec
        return norm_2(self._r_vec) if pre_op is not None else math.sqrt(self._gamma)


class BiCgStabSolver(IterativeSolver):
    """SolverBiCgStab.hpp:52-167 (unpreconditioned and right/left preconditioned branches)."""

    _native = "storm_hip_solve_bicgstab"

    def init(self, x_vec, b_vec, lin_op, pre_op):  # :59-91
        left_pre = pre_op is not None and self.pre_side == PreconditionerSide.Left
        names = ["_p_vec", "_r_vec", "_r_tilde_vec", "_t_vec", "_v_vec"] + (["_z_vec"] if pre_op is not None else [])
        for nme in names:
            v = DeviceVector()
            v.assign(x_vec, False)
            setattr(self, nme, v)
        lin_op.Residual(self._r_vec, b_vec, x_vec)
        if left_pre:
            self._z_vec, self._r_vec = self._r_vec, self._z_vec
            pre_op.mul(self._r_vec, self._z_vec)
        self._r_tilde_vec <<= self._r_vec
        self._rho = dot_product(self._r_tilde_vec, self._r_vec)
        self._alpha = self._omega = 0.0
        return math.sqrt(self._rho)

    def iterate(self, x_vec, b_vec, lin_op, pre_op):  # :93-165
        left_pre = pre_op is not None and self.pre_side == PreconditionerSide.Left
        right_pre = pre_op is not None and self.pre_side == PreconditionerSide.Right
        if self.iteration == 0:
            self._p_vec <<= self._r_vec
        else:
            rho_bar, self._rho = self._rho, dot_product(self._r_tilde_vec, self._r_vec)
            beta = safe_divide(self._alpha * self._rho, self._omega * rho_bar)
            self._p_vec <<= self._r_vec + beta * (self._p_vec - self._omega * self._v_vec)
        if left_pre:
            pre_op.mul_chain(self._v_vec, self._z_vec, lin_op, self._p_vec)
        elif right_pre:
            lin_op.mul_chain(self._v_vec, self._z_vec, pre_op, self._p_vec)
        else:
            lin_op.mul(self._v_vec, self._p_vec)
        self._alpha = safe_divide(self._rho, dot_product(self._r_tilde_vec, self._v_vec))
        x_vec += self._alpha * (self._z_vec if right_pre else self._p_vec)
        self._r_vec -= self._alpha * self._v_vec
        if left_pre:
            pre_op.mul_chain(self._t_vec, self._z_vec, lin_op, self._r_vec)
        elif right_pre:
            lin_op.mul_chain(self._t_vec, self._z_vec, pre_op, self._r_vec)
        else:
            lin_op.mul(self._t_vec, self._r_vec)
        self._omega = safe_divide(dot_product(self._t_vec, self._r_vec), dot_product(self._t_vec, self._t_vec))
        x_vec += self._omega * (self._z_vec if right_pre else self._r_vec)
        self._r_vec -= self._omega * self._t_vec
        return norm_2(self._r_vec)


class GmresSolver(InnerOuterIterativeSolver):
    """SolverGmres.hpp:41-255 (``BaseGmresSolver<Vector, Flexible>``): the unpreconditioned branches run
    natively on the device; with a ``pre_op`` the statement-level path below takes the reference's
    left / right (``Flexible = false``) or flexible (always right, :98-99) branches."""

    _native = "storm_hip_solve_gmres"
    _flexible = False

    def __init__(self):
        super().__init__()
        self.gram_schmidt = 0  # 0: modified (reference); 1: classical x2 (batched reductions)

    def _params(self):
        p = super()._params()
        p.gram_schmidt = self.gram_schmidt
        return p

    def _left_pre(self, pre_op) -> bool:
        return pre_op is not None and not self._flexible and self.pre_side == PreconditionerSide.Left

    def _right_pre(self, pre_op) -> bool:
        return pre_op is not None and (self._flexible or self.pre_side == PreconditionerSide.Right)

    def _start(self, x_vec, b_vec, lin_op, pre_op):  # :66-90 == :93-117
        q, z = self._q_vecs, self._z_vecs
        lin_op.Residual(q[0], b_vec, x_vec)
        if self._left_pre(pre_op):
            z[0], q[0] = q[0], z[0]
            pre_op.mul(q[0], z[0])
        self._beta[0] = norm_2(q[0])
        q[0] /= self._beta[0]

    def outer_init(self, x_vec, b_vec, lin_op, pre_op):  # :51-91
        m = self.num_inner_iterations
        self._beta = np.zeros(m + 1)
        self._cs, self._sn = np.zeros(m), np.zeros(m)
        self._H = np.zeros((m + 1, m))
        self._q_vecs: List[DeviceVector] = [_like(x_vec) for _ in range(m + 1)]
        self._z_vecs: List[DeviceVector] = []
        if pre_op is not None:  # :62-65
            self._z_vecs = [_like(x_vec) for _ in range(m if self._flexible else 1)]
        self._start(x_vec, b_vec, lin_op, pre_op)
        return self._beta[0]

    def inner_init(self, x_vec, b_vec, lin_op, pre_op):  # :93-117
        self._start(x_vec, b_vec, lin_op, pre_op)

    def inner_iterate(self, x_vec, b_vec, lin_op, pre_op):  # :119-192
        k, H, q, z = self.inner_iteration, self._H, self._q_vecs, self._z_vecs
        if self._left_pre(pre_op):  # :148-149
            pre_op.mul_chain(q[k + 1], z[0], lin_op, q[k])
        elif self._right_pre(pre_op):  # :150-152
            j = k if self._flexible else 0
            lin_op.mul_chain(q[k + 1], z[j], pre_op, q[k])
        else:
            lin_op.mul(q[k + 1], q[k])
        for i in range(k + 1):
            H[i, k] = dot_product(q[k + 1], q[i])
            q[k + 1] -= H[i, k] * q[i]
        H[k + 1, k] = norm_2(q[k + 1])
        q[k + 1] /= H[k + 1, k]
        cs, sn, beta = self._cs, self._sn, self._beta
        for i in range(k):
            chi = cs[i] * H[i, k] + sn[i] * H[i + 1, k]
            H[i + 1, k] = -sn[i] * H[i, k] + cs[i] * H[i + 1, k]
            H[i, k] = chi
        cs[k], sn[k], _ = sym_ortho(H[k, k], H[k + 1, k])
        H[k, k] = cs[k] * H[k, k] + sn[k] * H[k + 1, k]
        H[k + 1, k] = 0.0
        beta[k + 1] = -sn[k] * beta[k]
        beta[k] *= cs[k]
        return abs(beta[k + 1])

    def inner_finalize(self, x_vec, b_vec, lin_op, pre_op):  # :194-249
        k, H, beta, q, z = self.inner_iteration, self._H, self._beta, self._q_vecs, self._z_vecs
        for i in range(k, -1, -1):
            for j in range(i + 1, k + 1):
                beta[i] -= H[i, j] * beta[j]
            beta[i] /= H[i, i]
        if not self._right_pre(pre_op):  # :233-236
            for i in range(k + 1):
                x_vec += beta[i] * q[i]
        elif self._flexible:  # :237-240
            for i in range(k + 1):
                x_vec += beta[i] * z[i]
        else:  # :241-247
            q[0] *= beta[0]
            for i in range(1, k + 1):
                q[0] += beta[i] * q[i]
            pre_op.mul(z[0], q[0])
            x_vec += z[0]


class RichardsonSolver(IterativeSolver):
    """SolverRichardson.hpp:41-98 (x += omega r with the fixed ``relaxation_factor``)."""

    def __init__(self):
        super().__init__()
        self.relaxation_factor = 1.0e-4  # :45

    def _apply_pre(self, pre_op):
        if pre_op is not None:
            self._z_vec, self._r_vec = self._r_vec, self._z_vec
            pre_op.mul(self._r_vec, self._z_vec)

    def init(self, x_vec, b_vec, lin_op, pre_op):
        self._r_vec, self._z_vec = DeviceVector(), DeviceVector()
        self._r_vec.assign(x_vec, False)
        if pre_op is not None:
            self._z_vec.assign(x_vec, False)
        lin_op.Residual(self._r_vec, b_vec, x_vec)
        self._apply_pre(pre_op)
        return norm_2(self._r_vec)

    def iterate(self, x_vec, b_vec, lin_op, pre_op):
        x_vec += self.relaxation_factor * self._r_vec
        lin_op.Residual(self._r_vec, b_vec, x_vec)
        self._apply_pre(pre_op)
        return norm_2(self._r_vec)


def _side_mul(solver, pre_op, lin_op, z_vec, y_vec, x_vec) -> None:
    """The three-way dispatch every preconditioned solver body repeats (e.g. SolverBiCgStab.hpp:134-137):
    left ``z = P(y = A x)``, right ``z = A(y = P x)``, otherwise ``z = A x``."""
    if pre_op is not None and solver.pre_side == PreconditionerSide.Left:
        pre_op.mul_chain(z_vec, y_vec, lin_op, x_vec)
    elif pre_op is not None and solver.pre_side == PreconditionerSide.Right:
        lin_op.mul_chain(z_vec, y_vec, pre_op, x_vec)
    else:
        lin_op.mul(z_vec, x_vec)


class CgsSolver(IterativeSolver):
    """SolverCgs.hpp:50-176 (unpreconditioned, left and right preconditioned branches)."""

    def init(self, x_vec, b_vec, lin_op, pre_op):  # :57-90
        left_pre = pre_op is not None and self.pre_side == PreconditionerSide.Left
        for nme in ("_p_vec", "_q_vec", "_r_vec", "_r_tilde_vec", "_u_vec", "_v_vec"):
            setattr(self, nme, _like(x_vec))
        lin_op.Residual(self._r_vec, b_vec, x_vec)
        if left_pre:  # :81-84
            self._u_vec, self._r_vec = self._r_vec, self._u_vec
            pre_op.mul(self._r_vec, self._u_vec)
        self._r_tilde_vec <<= self._r_vec
        self._rho = dot_product(self._r_tilde_vec, self._r_vec)
        return math.sqrt(self._rho)

    def iterate(self, x_vec, b_vec, lin_op, pre_op):  # :92-174
        left_pre = pre_op is not None and self.pre_side == PreconditionerSide.Left
        right_pre = pre_op is not None and self.pre_side == PreconditionerSide.Right
        p, q, r, u, v = self._p_vec, self._q_vec, self._r_vec, self._u_vec, self._v_vec
        if self.iteration == 0:
            u <<= r
            p <<= u
        else:
            rho_bar, self._rho = self._rho, dot_product(self._r_tilde_vec, r)
            beta = safe_divide(self._rho, rho_bar)
            u <<= r + beta * q
            p <<= u + beta * (q + beta * p)
        _side_mul(self, pre_op, lin_op, v, q, p)  # :137-139
        alpha = safe_divide(self._rho, dot_product(self._r_tilde_vec, v))
        q <<= u - alpha * v
        v <<= u + q
        if left_pre:  # :159-162
            x_vec += alpha * v
            pre_op.mul_chain(v, u, lin_op, v)
            r -= alpha * v
        elif right_pre:  # :163-166
            lin_op.mul_chain(v, u, pre_op, v)
            x_vec += alpha * u
            r -= alpha * v
        else:  # :167-169
            lin_op.mul(u, v)
            x_vec += alpha * v
            r -= alpha * u
        return norm_2(r)


class _BaseTfqmrSolver(IterativeSolver):
    """SolverTfqmr.hpp:37-206 (unpreconditioned, left and right preconditioned); ``_L1`` selects TFQMR1."""

    _L1 = False

    def init(self, x_vec, b_vec, lin_op, pre_op):  # :45-88
        left_pre = pre_op is not None and self.pre_side == PreconditionerSide.Left
        for nme in ("_d_vec", "_r_tilde_vec", "_u_vec", "_v_vec", "_y_vec", "_s_vec"):
            setattr(self, nme, _like(x_vec))
        self._z_vec = _like(x_vec) if pre_op is not None else None
        if self._L1:
            self._d_vec <<= x_vec
        else:
            fill_with(self._d_vec, 0.0)
        lin_op.Residual(self._y_vec, b_vec, x_vec)
        if left_pre:  # :79-82
            self._z_vec, self._y_vec = self._y_vec, self._z_vec
            pre_op.mul(self._y_vec, self._z_vec)
        self._u_vec <<= self._y_vec
        self._r_tilde_vec <<= self._u_vec
        self._rho = dot_product(self._r_tilde_vec, self._u_vec)
        self._tau = math.sqrt(self._rho)
        return self._tau

    def iterate(self, x_vec, b_vec, lin_op, pre_op):  # :90-206
        right_pre = pre_op is not None and self.pre_side == PreconditionerSide.Right
        d, u, v, y, s, z = self._d_vec, self._u_vec, self._v_vec, self._y_vec, self._s_vec, self._z_vec
        if self.iteration == 0:
            _side_mul(self, pre_op, lin_op, s, z, y)
            v <<= s
        else:
            rho_bar, self._rho = self._rho, dot_product(self._r_tilde_vec, u)
            beta = safe_divide(self._rho, rho_bar)
            v <<= s + beta * v
            y <<= u + beta * y
            _side_mul(self, pre_op, lin_op, s, z, y)
            v <<= s + beta * v
        alpha = safe_divide(self._rho, dot_product(self._r_tilde_vec, v))
        for m in range(2):
            u -= alpha * s
            d += alpha * (z if right_pre else y)  # :169
            omega = norm_2(u)
            if self._L1:
                if omega < self._tau:
                    self._tau = omega
                    x_vec <<= d
            else:
                cs, sn, _ = sym_ortho(self._tau, omega)
                self._tau = omega * cs
                x_vec += (cs ** 2) * d
                d *= sn ** 2
            if m == 0:
                y -= alpha * v
                _side_mul(self, pre_op, lin_op, s, z, y)
        tau_tilde = self._tau
        if not self._L1:
            tau_tilde *= math.sqrt(2.0 * self.iteration + 3.0)
        return tau_tilde


class TfqmrSolver(_BaseTfqmrSolver):
    """SolverTfqmr.hpp:235-237."""


class Tfqmr1Solver(_BaseTfqmrSolver):
    """SolverTfqmr.hpp:262-264."""

    _L1 = True


class FgmresSolver(GmresSolver):
    """SolverGmres.hpp:306-308: flexible GMRES -- keeps every preconditioned vector z_k so the
    preconditioner may change between iterations; right preconditioning only.  Without a
    preconditioner it is GMRES and runs natively."""

    _flexible = True


class NewtonSolver(IterativeSolver):
    """SolverNewton.hpp:55-72: declared but unimplemented in the reference (``STORM_ABORT``)."""

    def init(self, x_vec, b_vec, any_op, pre_op):
        raise NotImplementedError("Newton solver is not implemented yet!")  # :61

    def iterate(self, x_vec, b_vec, any_op, pre_op):
        raise NotImplementedError("Newton solver is not implemented yet!")  # :67


class JfnkSolver(IterativeSolver):
    """SolverNewton.hpp:101-173: first-order Jacobian-free Newton-Krylov.  ``any_op`` may be nonlinear;
    each iteration solves ``J(x) t = r`` with a BiCGStab (1e-8 tolerances, :133-135) on the
    finite-difference Jacobian-vector product ``(A(x + delta y) - A(x)) / delta`` (:136-148)."""

    def init(self, x_vec, b_vec, any_op, pre_op):  # :106-122
        self._s_vec, self._t_vec, self._r_vec, self._w_vec = (_like(x_vec) for _ in range(4))
        self.inner_iterations = 0
        any_op.mul(self._w_vec, x_vec)
        self._r_vec <<= b_vec - self._w_vec
        return norm_2(self._r_vec)

    def iterate(self, x_vec, b_vec, any_op, pre_op):  # :124-161
        mu = math.sqrt(np.finfo(np.float64).eps) * math.sqrt(1.0 + norm_2(x_vec))
        self._t_vec <<= self._r_vec
        solver = BiCgStabSolver()
        solver.absolute_error_tolerance = 1.0e-8
        solver.relative_error_tolerance = 1.0e-8

        def jacobian_vector_product(z_vec, y_vec):  # :136-148
            delta = safe_divide(mu, norm_2(y_vec))
            self._s_vec <<= x_vec + delta * y_vec
            any_op.mul(z_vec, self._s_vec)
            delta_inverse = safe_divide(1.0, delta)
            z_vec <<= delta_inverse * (z_vec - self._w_vec)

        solver.solve(self._t_vec, self._r_vec, make_operator(jacobian_vector_product))
        self.inner_iterations += solver.iteration
        x_vec += self._t_vec
        any_op.mul(self._w_vec, x_vec)
        self._r_vec <<= b_vec - self._w_vec
        return norm_2(self._r_vec)


class BiCgStabLSolver(InnerOuterIterativeSolver):
    """SolverBiCgStab.hpp:184-383; ``num_inner_iterations`` is l (default 2).  A preconditioner is always
    applied on the left, whatever ``pre_side`` says (:226-229, :273-274, :292-293)."""

    def __init__(self):
        super().__init__()
        self.num_inner_iterations = 2  # :379-381

    def outer_init(self, x_vec, b_vec, lin_op, pre_op):
        l = self.num_inner_iterations
        self._gamma, self._gamma_bar = np.zeros(l + 1), np.zeros(l + 1)
        self._gamma_bbar, self._sigma = np.zeros(l + 1), np.zeros(l + 1)
        self._tau = np.zeros((l + 1, l + 1))
        mk = lambda: _like(x_vec)  # noqa: E731
        self._r_tilde_vec = mk()
        self._r_vecs = [mk() for _ in range(l + 1)]
        self._u_vecs = [mk() for _ in range(l + 1)]
        self._z_vec = mk() if pre_op is not None else None
        fill_with(self._u_vecs[0], 0.0)
        lin_op.Residual(self._r_vecs[0], b_vec, x_vec)
        if pre_op is not None:  # :226-229
            self._z_vec, self._r_vecs[0] = self._r_vecs[0], self._z_vec
            pre_op.mul(self._r_vecs[0], self._z_vec)
        self._r_tilde_vec <<= self._r_vecs[0]
        self._rho = dot_product(self._r_tilde_vec, self._r_vecs[0])
        self._alpha = self._omega = 0.0
        return math.sqrt(self._rho)

    def inner_iterate(self, x_vec, b_vec, lin_op, pre_op):
        l, j = self.num_inner_iterations, self.inner_iteration
        r, u = self._r_vecs, self._u_vecs
        if self.iteration == 0:
            u[0] <<= r[0]
        else:
            rho_bar, self._rho = self._rho, dot_product(self._r_tilde_vec, r[j])
            beta = safe_divide(self._alpha * self._rho, rho_bar)
            for i in range(j + 1):
                u[i] <<= r[i] - beta * u[i]
        if pre_op is not None:
            pre_op.mul_chain(u[j + 1], self._z_vec, lin_op, u[j])
        else:
            lin_op.mul(u[j + 1], u[j])
        self._alpha = safe_divide(self._rho, dot_product(self._r_tilde_vec, u[j + 1]))
        for i in range(j + 1):
            r[i] -= self._alpha * u[i + 1]
        x_vec += self._alpha * u[0]
        if pre_op is not None:
            pre_op.mul_chain(r[j + 1], self._z_vec, lin_op, r[j])
        else:
            lin_op.mul(r[j + 1], r[j])
        if j == l - 1:
            tau, sigma, g, gb, gbb = self._tau, self._sigma, self._gamma, self._gamma_bar, self._gamma_bbar
            for jj in range(1, l + 1):
                for i in range(1, jj):
                    tau[i, jj] = safe_divide(dot_product(r[i], r[jj]), sigma[i])
                    r[jj] -= tau[i, jj] * r[i]
                sigma[jj] = dot_product(r[jj], r[jj])
                gb[jj] = safe_divide(dot_product(r[0], r[jj]), sigma[jj])
            self._omega = g[l] = gb[l]
            self._rho *= -self._omega
            for jj in range(l - 1, 0, -1):
                g[jj] = gb[jj]
                for i in range(jj + 1, l + 1):
                    g[jj] -= tau[jj, i] * g[i]
            for jj in range(1, l):
                gbb[jj] = g[jj + 1]
                for i in range(jj + 1, l):
                    gbb[jj] += tau[jj, i] * g[i + 1]
            x_vec += g[1] * r[0]
            r[0] -= gb[l] * r[l]
            u[0] -= g[l] * u[l]
            for jj in range(1, l):
                x_vec += gbb[jj] * r[jj]
                r[0] -= gb[jj] * r[jj]
                u[0] -= g[jj] * u[jj]
        return norm_2(r[0])


class IdrsSolver(InnerOuterIterativeSolver):
    """SolverIdrs.hpp:52-291 (unpreconditioned, left and right preconditioned); ``num_inner_iterations`` is s
    (default 4)."""

    def __init__(self):
        super().__init__()
        self.num_inner_iterations = 4  # :287-289

    def outer_init(self, x_vec, b_vec, lin_op, pre_op):
        s = self.num_inner_iterations
        self._phi, self._gamma, self._mu = np.zeros(s), np.zeros(s), np.zeros((s, s))
        mk = lambda: _like(x_vec)  # noqa: E731
        self._r_vec, self._v_vec = mk(), mk()
        self._p_vecs = [mk() for _ in range(s)]
        self._u_vecs = [mk() for _ in range(s)]
        self._g_vecs = [mk() for _ in range(s)]
        self._z_vec = mk() if pre_op is not None else None
        lin_op.Residual(self._r_vec, b_vec, x_vec)
        if pre_op is not None and self.pre_side == PreconditionerSide.Left:  # :100-103
            self._z_vec, self._r_vec = self._r_vec, self._z_vec
            pre_op.mul(self._r_vec, self._z_vec)
        self._phi[0] = norm_2(self._r_vec)
        return self._phi[0]

    def inner_init(self, x_vec, b_vec, lin_op, pre_op):
        s, p, phi, mu = self.num_inner_iterations, self._p_vecs, self._phi, self._mu
        if self.iteration == 0:
            self._omega = mu[0, 0] = 1.0
            p[0] <<= self._r_vec / phi[0]
            for i in range(1, s):
                mu[i, i], phi[i] = 1.0, 0.0
                fill_randomly(p[i])
                for j in range(i):
                    mu[i, j] = 0.0
                    p[i] -= dot_product(p[i], p[j]) * p[j]
                p[i] /= norm_2(p[i])
        else:
            for i in range(s):
                phi[i] = dot_product(p[i], self._r_vec)

    def inner_iterate(self, x_vec, b_vec, lin_op, pre_op):
        s, k = self.num_inner_iterations, self.inner_iteration
        phi, gamma, mu = self._phi, self._gamma, self._mu
        left_pre = pre_op is not None and self.pre_side == PreconditionerSide.Left
        right_pre = pre_op is not None and self.pre_side == PreconditionerSide.Right
        p, u, g, r, v = self._p_vecs, self._u_vecs, self._g_vecs, self._r_vec, self._v_vec
        for i in range(k, s):
            gamma[i] = phi[i]
            for j in range(k, i):
                gamma[i] -= mu[i, j] * gamma[j]
            gamma[i] /= mu[i, i]
        v <<= r - gamma[k] * g[k]
        for i in range(k + 1, s):
            v -= gamma[i] * g[i]
        if right_pre:  # :204-207
            self._z_vec, self._v_vec = self._v_vec, self._z_vec
            v = self._v_vec
            pre_op.mul(v, self._z_vec)
        u[k] <<= self._omega * v + gamma[k] * u[k]
        for i in range(k + 1, s):
            u[k] += gamma[i] * u[i]
        if left_pre:  # :212-213
            pre_op.mul_chain(g[k], self._z_vec, lin_op, u[k])
        else:
            lin_op.mul(g[k], u[k])
        for i in range(k):
            alpha = safe_divide(dot_product(p[i], g[k]), mu[i, i])
            u[k] -= alpha * u[i]
            g[k] -= alpha * g[i]
        for i in range(k, s):
            mu[i, k] = dot_product(p[i], g[k])
        beta = safe_divide(phi[k], mu[k, k])
        x_vec += beta * u[k]
        r -= beta * g[k]
        for i in range(k + 1, s):
            phi[i] -= beta * mu[i, k]
        if k == s - 1:  # :268-279
            _side_mul(self, pre_op, lin_op, v, self._z_vec, r)
            self._omega = safe_divide(dot_product(v, r), dot_product(v, v))
            x_vec += self._omega * (self._z_vec if right_pre else r)
            r -= self._omega * v
        return norm_2(r)


def _like(v: DeviceVector) -> DeviceVector:
    w = DeviceVector()
    w.assign(v, False)
    return w


def solve(solver_cls, x_vec: DeviceVector, b_vec: DeviceVector, any_op: Operator) -> bool:
    """``solve<Solver>(x, b, op)``  Solver.hpp:261-265."""
    return solver_cls().solve(x_vec, b_vec, any_op)
