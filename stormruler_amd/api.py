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
falling back to a host loop.  The solver classes are knob holders: their loops run inside
libstorm_hip.so (``storm_hip_krylov_*``, csrc/krylov.hip) for ANY operator -- a
:class:`HipStencilOperator` binds natively, a Python lambda through ``make_operator`` as a callback
that only enqueues its kernels -- with every scalar of the recurrences device-resident.
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

    def counter(self, key: str) -> int:
        """How many solves of this context took a given path ("resident_solves", "latency_solves",
        "throughput_solves", "engine_solves", "cg_fused_steps")."""
        v = C.c_int64()
        check(lib.storm_hip_ctx_get_counter(self._h, key.encode(), C.byref(v)))
        return v.value

    RCCL_PROFILE_KEYS = ("rccl_prof_exchanges", "rccl_prof_event_to_comm_ticks", "rccl_prof_pack_ticks",
                         "rccl_prof_sendrecv_ticks", "rccl_prof_unhidden_wait_ticks", "rccl_prof_resume_ticks",
                         "rccl_prof_allreduces", "rccl_prof_allreduce_ticks")

    def rccl_profile(self, iterations: int) -> dict:
        """What the solves since ``set_option("profile_comm", 1)`` spent in the RCCL transport's halo exchanges and
        all-reduces, from device timestamps (one-thread stamp kernels between the launches of the compute and the comm
        stream, csrc/comm.hip): microseconds per exchange / per all-reduce and per iteration."""
        c = {k: self.counter(k) for k in self.RCCL_PROFILE_KEYS}
        ne, na, it = max(c["rccl_prof_exchanges"], 1), max(c["rccl_prof_allreduces"], 1), max(iterations, 1)
        us = 0.01  # ticks of 10 ns
        return {
            "transport": "rccl", "iterations_covered": iterations,
            "halo_exchanges_per_iteration": c["rccl_prof_exchanges"] / it,
            "allreduces_per_iteration": c["rccl_prof_allreduces"] / it,
            "event_to_comm_stream_us_each": c["rccl_prof_event_to_comm_ticks"] * us / ne,
            "pack_us_each": c["rccl_prof_pack_ticks"] * us / ne,
            "sendrecv_us_each": c["rccl_prof_sendrecv_ticks"] * us / ne,
            "halo_unhidden_wait_us_each": c["rccl_prof_unhidden_wait_ticks"] * us / ne,
            "halo_done_to_boundary_rows_us_each": c["rccl_prof_resume_ticks"] * us / ne,
            "allreduce_us_each": c["rccl_prof_allreduce_ticks"] * us / na,
            "halo_unhidden_us_per_iteration": (c["rccl_prof_unhidden_wait_ticks"] + c["rccl_prof_resume_ticks"]) * us / it,
            "allreduce_us_per_iteration": c["rccl_prof_allreduce_ticks"] * us / it,
            "note": "device timestamps of an INSTRUMENTED solve (a one-thread stamp kernel in front of and behind every step, "
                    "~2 us each on its stream): event_to_comm_stream = the compute stream has the vector ready -> the comm "
                    "stream starts (cross-stream event); pack; sendrecv = the grouped ncclSend / ncclRecv; halo_unhidden_wait = "
                    "the exchange still running when the interior rows had ended; halo_done_to_boundary_rows = exchange and "
                    "interior rows both done -> the compute stream resumes (the second cross-stream event); allreduce = "
                    "stamp to stamp around ncclAllReduce on the compute stream.  What an iteration loses to the transport is "
                    "halo_unhidden_us_per_iteration + allreduce_us_per_iteration (+ the pack kernel where it runs on the "
                    "compute stream)"}

    def spmv_profile(self):
        """(launches, total_ms, min_ms) of the SpMV kernel since the last call (option profile_spmv)."""
        n, tot, mn = C.c_int64(), C.c_double(), C.c_double()
        check(lib.storm_hip_ctx_get_spmv_profile(self._h, C.byref(n), C.byref(tot), C.byref(mn)))
        return n.value, tot.value, mn.value

    def spmv_profile_samples(self, capacity: int = 1 << 16):
        """Milliseconds of every SpMV launch since the last call, in launch order (option profile_spmv)."""
        buf = (C.c_double * capacity)()
        n = C.c_int64()
        check(lib.storm_hip_ctx_get_spmv_profile_samples(self._h, buf, capacity, C.byref(n)))
        return np.array(buf[: min(n.value, capacity)], dtype=np.float64)

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

    def rccl_view(self) -> dict:
        """What RCCL itself reports for this context's communicators (``storm_hip_ctx_comm_rccl_view``): rank count, this
        rank's number and device as the halo communicator sees them, the reduction communicator's rank count, the HIP
        device and its PCI bus id.  Counts of 0: no RCCL communicator on this context."""
        v = [C.c_int() for _ in range(5)]
        bus = C.create_string_buffer(32)
        check(lib.storm_hip_ctx_comm_rccl_view(self._h, *[C.byref(x) for x in v], bus, 32))
        return {"nccl_comm_count": v[0].value, "nccl_user_rank": v[1].value, "nccl_device": v[2].value,
                "reduction_comm_count": v[3].value, "hip_device": v[4].value, "pci_bus_id": bus.value.decode()}

    def comm_ipc_export(self, n_ranks: int, rank: int, window_bytes: int = 0) -> bytes:
        """Allocate this rank's peer window (``storm_hip_ctx_comm_ipc_export``); returns its 64-byte IPC handle."""
        buf = C.create_string_buffer(64)
        check(lib.storm_hip_ctx_comm_ipc_export(self._h, n_ranks, rank, int(window_bytes), buf))
        self.n_ranks, self.rank = n_ranks, rank
        return buf.raw

    def comm_init_ipc(self, handles: bytes):
        """Map every rank's window (``handles`` = the n_ranks exported handles, rank order)."""
        assert len(handles) == 64 * self.n_ranks
        check(lib.storm_hip_ctx_comm_init_ipc(self._h, C.create_string_buffer(handles, len(handles))))

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


# ``out <<= map(func, mats...)`` (Bittern/MatrixMath.hpp:44-105; the playground's ``f <<= map(dF_dc, c)``,
# Playground.cpp:142-148).  ``func`` runs on the device, so it is TRACED: called once with :class:`Sym` operands it
# records its arithmetic, in order, as a reverse-Polish program for ``storm_hip_map`` -- every operation rounded on its
# own, so the device's values equal numpy's evaluation of the same expression bit for bit.
MAP_X0, MAP_X1, MAP_Y, MAP_CONST, MAP_NEG, MAP_ABS, MAP_SQRT = 0, 1, 2, 3, 8, 9, 10
MAP_ADD, MAP_SUB, MAP_MUL, MAP_DIV, MAP_MIN, MAP_MAX = 16, 17, 18, 19, 20, 21


class Sym:
    """A traced scalar of an element map: supports + - * /, unary -, abs, ``sqrt()``, ``min()`` / ``max()`` with
    floats and other traced values.  No truth value (no data-dependent branches)."""

    __array_ufunc__ = None

    def __init__(self, code, consts):
        self.code, self.consts = list(code), list(consts)

    @staticmethod
    def of(v) -> "Sym":
        return v if isinstance(v, Sym) else Sym([MAP_CONST], [float(v)])

    def _bin(self, op, other, swap=False):
        a, b = (Sym.of(other), self) if swap else (self, Sym.of(other))
        code, consts = list(a.code), list(a.consts)
        for word in b.code:
            if word & 0xff == MAP_CONST:
                v = b.consts[word >> 8]
                for k, have in enumerate(consts):  # one slot per distinct bit pattern
                    if np.float64(have).tobytes() == np.float64(v).tobytes():
                        break
                else:
                    consts.append(v)
                    k = len(consts) - 1
                word = MAP_CONST | (k << 8)
            code.append(word)
        return Sym(code + [op], consts)

    def __add__(self, o): return self._bin(MAP_ADD, o)
    def __radd__(self, o): return self._bin(MAP_ADD, o, True)
    def __sub__(self, o): return self._bin(MAP_SUB, o)
    def __rsub__(self, o): return self._bin(MAP_SUB, o, True)
    def __mul__(self, o): return self._bin(MAP_MUL, o)
    def __rmul__(self, o): return self._bin(MAP_MUL, o, True)
    def __truediv__(self, o): return self._bin(MAP_DIV, o)
    def __rtruediv__(self, o): return self._bin(MAP_DIV, o, True)
    def __neg__(self): return Sym(self.code + [MAP_NEG], self.consts)
    def __pos__(self): return self
    def __abs__(self): return Sym(self.code + [MAP_ABS], self.consts)
    def sqrt(self): return Sym(self.code + [MAP_SQRT], self.consts)
    def min(self, o): return self._bin(MAP_MIN, o)
    def max(self, o): return self._bin(MAP_MAX, o)

    def __bool__(self):
        raise TypeError("a traced element has no truth value: an element map cannot branch on its data")


class _Mapped:  # map(func, x0 [, x1 [, x2]])
    def __init__(self, program: Sym, xs):
        self.program, self.xs = program, list(xs)


def map(func, *xs: "DeviceVector") -> _Mapped:  # noqa: A001 -- the reference's name (Bittern/MatrixMath.hpp:103)
    """``map(func, mats...)``: consumed by ``out <<= ...``.  One or two vectors, or three when one is the target."""
    if not 1 <= len(xs) <= 3:
        raise ValueError("map: one to three vector operands")
    return _Mapped(Sym.of(func(*[Sym([k], []) for k in range(len(xs))])), xs)


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

    @classmethod
    def _borrow(cls, ctx: Context, handle) -> "DeviceVector":
        """A non-owning view of a vector the library hands to an operator / preconditioner callback."""
        v = cls()
        v.ctx, v._h, v._owned = ctx, C.c_void_p(handle), False
        n, nh = C.c_int64(), C.c_int64()
        check(lib.storm_hip_vec_size(v._h, C.byref(n), C.byref(nh)))
        v.n_owned, v.n_halo = n.value, nh.value
        return v

    def _free(self):
        if getattr(self, "_h", None):
            # (a context closed explicitly takes its children with it; an object the garbage collector was
            #  already tearing down then is not in the weak set any more -- never touch a dead context)
            if getattr(self, "_owned", True) and getattr(self.ctx, "_h", None):
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
        elif isinstance(e, _Mapped):
            # the program's slots -> the kernel's operands: the target itself is `y`, the others x0, x1 in their order
            opcode_of, others = {}, []
            for k, x in enumerate(e.xs):
                if x is self:
                    opcode_of[k] = MAP_Y
                else:
                    if len(others) == 2:
                        raise ValueError("map: three operands, none of which is the target of `<<=`")
                    opcode_of[k] = MAP_X0 if not others else MAP_X1
                    others.append(x)
            code = [opcode_of[w] if (w & 0xff) < MAP_CONST else w for w in e.program.code]
            prog = (C.c_int32 * len(code))(*code)
            consts = (C.c_double * max(len(e.program.consts), 1))(*e.program.consts)
            check(lib.storm_hip_map(self._h, others[0]._h if others else None, others[1]._h if len(others) > 1 else None,
                                    prog, len(code), consts, len(e.program.consts)))
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


def vdiv(y: DeviceVector, s: float, a: Optional[DeviceVector], b: DeviceVector) -> None:
    """``y = (s * a) / b`` elementwise; ``a is None``: ``y = s / b`` (MatrixMath.hpp:261-265, :298-302)."""
    check(lib.storm_hip_vdiv(y._h, float(s), None if a is None else a._h, b._h))


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


class PendingDots:
    """``<a, bs[i]>`` enqueued, not yet awaited (``storm_hip_multi_dot_begin``): up to 8 in flight per context;
    ``result()`` waits for the sums (once)."""

    def __init__(self, a: DeviceVector, bs: Sequence[DeviceVector]):
        self._ctx, self._k = a.ctx, len(bs)
        arr = (C.c_void_p * self._k)(*[b._h for b in bs])
        req = C.c_int(0)
        check(lib.storm_hip_multi_dot_begin(a._h, arr, self._k, C.byref(req)))
        self._req = req.value

    def result(self) -> np.ndarray:
        out = np.empty(self._k)
        check(lib.storm_hip_multi_dot_end(self._ctx._h, self._req, out.ctypes.data_as(_lib.f64p)))
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
        """Diffusion stencil of ``stormDivGrad`` (Playground.cpp:115-131) from mesh quantities: the library forms the
        face transmissibilities A_f / d_f itself (``storm_hip_op_create_from_mesh``; the same bits as
        ``mesh.face_coefficients`` + ``storm_hip_op_create_from_faces``, see ``from_face_coefficients``)."""
        h = C.c_void_p()
        i64, f64 = _lib.i64p, _lib.f64p
        inner = np.ascontiguousarray(g.inner, np.int64)
        outer = np.ascontiguousarray(g.outer, np.int64)
        b_cell = np.ascontiguousarray(g.b_cell, np.int64)
        vol = np.ascontiguousarray(g.volume, np.float64)
        area = np.ascontiguousarray(g.area, np.float64)
        center = np.ascontiguousarray(g.center, np.float64)
        b_area = np.ascontiguousarray(g.b_area, np.float64)
        b_center = np.ascontiguousarray(g.b_center, np.float64)
        check(lib.storm_hip_op_create_from_mesh(
            ctx._h, g.n_cells, g.n_halo, g.dim, g.n_faces, inner.ctypes.data_as(i64), outer.ctypes.data_as(i64),
            area.ctypes.data_as(f64), center.ctypes.data_as(f64), g.n_bfaces, b_cell.ctypes.data_as(i64),
            b_area.ctypes.data_as(f64), b_center.ctypes.data_as(f64), vol.ctypes.data_as(f64), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_face_coefficients(cls, ctx: Context, g: FaceGraph) -> "StencilMatrix":
        """The same operator with A_f / d_f computed on the host (``mesh.face_coefficients``) and handed to
        ``storm_hip_op_create_from_faces``."""
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
            if getattr(self.ctx, "_h", None):  # never touch a dead context (see DeviceVector._free)
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


class _Engine:
    """One ``storm_hip_krylov`` object (the library's device-resident solver loops, csrc/krylov.hip) with the
    reference-side objects bound to it: a :class:`HipStencilOperator` binds natively, any other ``Operator`` --
    e.g. a lambda through ``make_operator``, the reference's only call site (Playground.cpp:151-167) -- as a
    callback that merely enqueues its kernels; likewise ``pre_op`` (a :class:`JacobiPreconditioner` binds as a
    device diagonal).  An exception raised inside a callback aborts the solve and is re-raised from it."""

    def __init__(self, ctx: Context, method: int):
        h = C.c_void_p()
        check(lib.storm_hip_krylov_create(ctx._h, method, C.byref(h)))
        self.ctx, self._h, self._keep, self._error = ctx, h, [], None
        ctx._children.add(self)

    def _free(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):
                lib.storm_hip_krylov_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self._free()
        except Exception:
            pass

    def _callback(self, fn):
        def trampoline(_user, y_handle, x_handle):
            try:
                fn(DeviceVector._borrow(self.ctx, y_handle), DeviceVector._borrow(self.ctx, x_handle))
                return 0
            except BaseException as e:  # never unwind through the C frames
                self._error = e
                return 1

        cb = _lib.APPLY_FN(trampoline)
        self._keep.append(cb)
        return cb

    def bind(self, any_op: Operator, pre_op, pre_side: int) -> None:
        self._keep, self._error = [], None
        if isinstance(any_op, HipStencilOperator):
            check(lib.storm_hip_krylov_set_operator(self._h, any_op.matrix._h, any_op.alpha, any_op.beta))
        else:
            check(lib.storm_hip_krylov_set_operator_fn(self._h, self._callback(any_op.mul), None))
        if pre_op is None:
            check(lib.storm_hip_krylov_set_preconditioner_diag(self._h, None, int(pre_side)))
        elif isinstance(pre_op, JacobiPreconditioner):
            check(lib.storm_hip_krylov_set_preconditioner_diag(self._h, pre_op._dinv._h, int(pre_side)))
        else:
            check(lib.storm_hip_krylov_set_preconditioner_fn(self._h, self._callback(pre_op.mul), None, int(pre_side)))

    def call(self, status: int) -> None:
        err, self._error = self._error, None
        if err is not None:
            raise err
        check(status)


class IterativeSolver(Solver):
    """Solver.hpp:62-149 -- same public knobs, defaults and convergence rule.

    The solvers this module ships set ``_method`` and run inside libstorm_hip.so (``storm_hip_krylov_solve``):
    the loop of Solver.hpp:132-140 is evaluated on the device, the host never waits for a scalar.  Their
    ``init`` / ``iterate`` / ``finalize`` -- the reference's protected stepping interface (:78-111) -- go
    through ``storm_hip_krylov_init / _iterate / _finalize`` (one host wait per call); ``device_loop = False``
    makes ``solve`` run the reference's host loop over them.  A subclass that sets no ``_method`` and
    overrides the three hooks gets that host loop as well."""

    _method: Optional[int] = None

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
        self.num_applies = 0        # operator applications of the last solve (logical: what the reference does)
        self.num_pre_applies = 0    # preconditioner applications, likewise
        self.initial_error = 0.0
        self.device_loop = True
        self.lazy_statements = True  # the host loop runs with the library's option of that name (csrc/lazy.hip)
        self._engine: Optional[_Engine] = None

    # -- the reference's stepping hooks ------------------------------------------------------
    def init(self, x_vec, b_vec, any_op, pre_op) -> float:
        if self._method is None:
            raise NotImplementedError
        eng = self._bound_engine(x_vec, any_op, pre_op)
        err = C.c_double()
        p = self._params()
        eng.call(lib.storm_hip_krylov_init(eng._h, b_vec._h, x_vec._h, C.byref(p), C.byref(err)))
        return err.value

    def iterate(self, x_vec, b_vec, any_op, pre_op) -> float:
        if self._method is None:
            raise NotImplementedError
        err = C.c_double()
        self._engine.call(lib.storm_hip_krylov_iterate(self._engine._h, C.byref(err)))
        return err.value

    def finalize(self, x_vec, b_vec, any_op, pre_op) -> None:
        if self._method is not None:
            self._engine.call(lib.storm_hip_krylov_finalize(self._engine._h))

    # -- plumbing ------------------------------------------------------------------------------
    def _params(self) -> _lib.SolverParams:
        p = _lib.SolverParams()
        lib.storm_hip_solver_params_default(C.byref(p))
        p.num_iterations = self.num_iterations
        p.absolute_error_tolerance = self.absolute_error_tolerance
        p.relative_error_tolerance = self.relative_error_tolerance
        p.check_lag = self.check_lag
        p.num_inner_iterations = int(getattr(self, "num_inner_iterations", 0))
        p.gram_schmidt = int(getattr(self, "gram_schmidt", 0))
        return p

    def _configure(self, eng: _Engine) -> None:
        pass

    def _bound_engine(self, x_vec, any_op, pre_op) -> _Engine:
        if self._engine is None or self._engine.ctx is not x_vec.ctx or self._engine._h is None:
            self._engine = _Engine(x_vec.ctx, self._method)
        self._engine.bind(any_op, pre_op, self.pre_side)
        self._configure(self._engine)
        return self._engine

    def _log(self) -> None:
        """The reference's one line per solve (`STORM_INFO`, Solver.hpp:144-145) on the logger
        ``stormruler_amd.solvers`` at INFO level."""
        _LOG.info("n_iter: %4d, abs_err: %-12e, rel_err: %-12e", self.iteration, self.absolute_error, self.relative_error)

    def solve(self, x_vec, b_vec, any_op) -> bool:  # Solver.hpp:116-147
        if self.pre_op is not None:
            self.pre_op.build(x_vec, b_vec, any_op)
        if self._method is not None and self.device_loop:
            eng = self._bound_engine(x_vec, any_op, self.pre_op)
            p, r, n_pre = self._params(), _lib.SolverResult(), C.c_int64()
            hist = np.zeros(self.num_iterations + 1) if self.record_history else None
            eng.call(lib.storm_hip_krylov_solve(eng._h, b_vec._h, x_vec._h, C.byref(p), C.byref(r),
                                                None if hist is None else hist.ctypes.data_as(_lib.f64p),
                                                C.byref(n_pre)))
            self.iteration = r.iterations
            self.absolute_error, self.relative_error = r.absolute_error, r.relative_error
            self.initial_error, self.num_applies, self.num_pre_applies = r.initial_error, r.num_applies, n_pre.value
            self.history = None if hist is None else hist[: r.iterations + 1]
            # 0, or why the library left the path it had chosen (storm_hip_solver_result::path_fallback): 1 = a
            # cooperative kernel could not be launched, 2 = one gave up waiting and the solve was re-run without
            self.path_fallback = r.path_fallback
            self._log()
            return bool(r.converged)
        # the host loop: user-defined solvers, and the shipped ones stepwise when device_loop is off.  It runs with the
        # library's option lazy_statements (csrc/lazy.hip): the body's linear statements and applies wait for the call that
        # needs them, and a reduction over what the last one writes rides in its kernel -- the same values, bit for bit
        lazy_ctx = x_vec.ctx if (self.lazy_statements and getattr(x_vec, "ctx", None) is not None) else None
        if lazy_ctx is not None:
            lazy_ctx.set_option("lazy_statements", 2)
        try:
            return self._host_loop(x_vec, b_vec, any_op)
        finally:
            if lazy_ctx is not None:
                lazy_ctx.set_option("lazy_statements", 0)

    def _host_loop(self, x_vec, b_vec, any_op) -> bool:
        initial_error = self.init(x_vec, b_vec, any_op, self.pre_op)
        self.initial_error = self.absolute_error = initial_error
        hist = [initial_error]
        self.iteration = 0
        converged = self.absolute_error_tolerance > 0.0 and initial_error < self.absolute_error_tolerance
        early = converged
        while not converged and self.iteration < self.num_iterations:
            self.absolute_error = self.iterate(x_vec, b_vec, any_op, self.pre_op)
            self.relative_error = self.absolute_error / initial_error
            hist.append(self.absolute_error)
            converged = ((self.absolute_error_tolerance > 0.0 and self.absolute_error < self.absolute_error_tolerance)
                         or (self.relative_error_tolerance > 0.0 and self.relative_error < self.relative_error_tolerance))
            self.iteration += 1
        self.finalize(x_vec, b_vec, any_op, self.pre_op)
        self.history = np.array(hist)
        if not early:
            self._log()
        return bool(converged)


class InnerOuterIterativeSolver(IterativeSolver):
    """Solver.hpp:154-259: ``num_inner_iterations`` (default 50) and the restart bookkeeping.  For the shipped
    solvers that bookkeeping (``iteration % num_inner_iterations``, inner_init at 0, inner_finalize at the end
    of a cycle and once more on exit) runs inside the library; a user-defined subclass implements the five
    ``outer_* / inner_*`` hooks and gets it from here."""

    def __init__(self):
        super().__init__()
        self.inner_iteration = 0
        self.num_inner_iterations = 50

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

    def init(self, x_vec, b_vec, any_op, pre_op):
        if self._method is not None:
            return super().init(x_vec, b_vec, any_op, pre_op)
        return self.outer_init(x_vec, b_vec, any_op, pre_op)

    def iterate(self, x_vec, b_vec, any_op, pre_op):
        self.inner_iteration = self.iteration % self.num_inner_iterations
        if self._method is not None:
            return super().iterate(x_vec, b_vec, any_op, pre_op)
        last = self.inner_iteration == self.num_inner_iterations - 1
        if self.inner_iteration == 0:
            self.inner_init(x_vec, b_vec, any_op, pre_op)
        residual_norm = self.inner_iterate(x_vec, b_vec, any_op, pre_op)
        if last:
            self.inner_finalize(x_vec, b_vec, any_op, pre_op)
        return residual_norm

    def finalize(self, x_vec, b_vec, any_op, pre_op):
        if self._method is not None:
            return super().finalize(x_vec, b_vec, any_op, pre_op)
        if self.inner_iteration != self.num_inner_iterations - 1:
            self.inner_finalize(x_vec, b_vec, any_op, pre_op)
        self.outer_finalize(x_vec, b_vec, any_op, pre_op)


# The shipped solvers: a method id each; their loops are csrc/krylov.hip (and, for a stencil operator without
# preconditioner, the fused kernels of csrc/solvers.hip).
class CgSolver(IterativeSolver):
    """SolverCg.hpp:47-128."""

    _method = 0  # STORM_HIP_CG


class BiCgStabSolver(IterativeSolver):
    """SolverBiCgStab.hpp:52-167 (unpreconditioned and left / right preconditioned branches)."""

    _method = 1  # STORM_HIP_BICGSTAB


class GmresSolver(InnerOuterIterativeSolver):
    """SolverGmres.hpp:41-255, :281-283.  ``gram_schmidt = 1`` switches the orthogonalisation from the
    reference's modified Gram-Schmidt to classical Gram-Schmidt applied twice (batched reductions)."""

    _method = 2  # STORM_HIP_GMRES

    def __init__(self):
        super().__init__()
        self.gram_schmidt = 0


class FgmresSolver(GmresSolver):
    """SolverGmres.hpp:306-308: flexible GMRES -- keeps every preconditioned vector z_k so the preconditioner
    may change between iterations; right preconditioning only."""

    _method = 3  # STORM_HIP_FGMRES


class CgsSolver(IterativeSolver):
    """SolverCgs.hpp:50-176."""

    _method = 4  # STORM_HIP_CGS


class TfqmrSolver(IterativeSolver):
    """SolverTfqmr.hpp:227-240 (L2 quasi-minimisation)."""

    _method = 5  # STORM_HIP_TFQMR


class Tfqmr1Solver(IterativeSolver):
    """SolverTfqmr.hpp:252-265 (L1 variant)."""

    _method = 6  # STORM_HIP_TFQMR1


class BiCgStabLSolver(InnerOuterIterativeSolver):
    """SolverBiCgStab.hpp:184-383; ``num_inner_iterations`` is l (default 2).  A preconditioner is always
    applied on the left, whatever ``pre_side`` says (:226-229, :273-274, :292-293)."""

    _method = 7  # STORM_HIP_BICGSTAB_L

    def __init__(self):
        super().__init__()
        self.num_inner_iterations = 2  # :379-381


class IdrsSolver(InnerOuterIterativeSolver):
    """SolverIdrs.hpp:52-291; ``num_inner_iterations`` is s (default 4).  The shadow space comes from
    ``fill_randomly`` (the reference's engine and sequence; ``rng_reset()`` = a fresh process)."""

    _method = 8  # STORM_HIP_IDRS

    def __init__(self):
        super().__init__()
        self.num_inner_iterations = 4  # :287-289


class RichardsonSolver(IterativeSolver):
    """SolverRichardson.hpp:41-98."""

    _method = 9  # STORM_HIP_RICHARDSON

    def __init__(self):
        super().__init__()
        self.relaxation_factor = 1.0e-4  # :45

    def _configure(self, eng: _Engine) -> None:
        check(lib.storm_hip_krylov_set_real(eng._h, b"relaxation_factor", float(self.relaxation_factor)))


class NewtonSolver(IterativeSolver):
    """SolverNewton.hpp:55-72: declared but unimplemented in the reference (``STORM_ABORT``)."""

    def init(self, x_vec, b_vec, any_op, pre_op):
        raise NotImplementedError("Newton solver is not implemented yet!")  # :61

    def iterate(self, x_vec, b_vec, any_op, pre_op):
        raise NotImplementedError("Newton solver is not implemented yet!")  # :67


class JfnkSolver(IterativeSolver):
    """SolverNewton.hpp:101-173: first-order Jacobian-free Newton-Krylov.  ``any_op`` may be nonlinear.  A
    user-level solver on the host loop above: every Newton step solves ``J(x) t = r`` with a
    :class:`BiCgStabSolver` at 1e-8 (:133-135) whose operator is the finite-difference product
    ``(A(x + delta y) - A(x)) / delta``, ``delta = mu / |y|`` (:136-148)."""

    def _linearised_at(self, x_vec, any_op) -> "FunctionalOperator":
        mu = math.sqrt(np.finfo(np.float64).eps) * math.sqrt(1.0 + norm_2(x_vec))  # :131
        shifted, at_x = self._work[0], self._work[3]

        def product(z_vec, y_vec):
            delta = safe_divide(mu, norm_2(y_vec))
            s_vec = shifted
            s_vec <<= x_vec + delta * y_vec
            any_op.mul(z_vec, s_vec)
            z_vec <<= safe_divide(1.0, delta) * (z_vec - at_x)

        return make_operator(product)

    def _residual(self, x_vec, b_vec, any_op) -> float:
        at_x, res = self._work[3], self._work[2]
        any_op.mul(at_x, x_vec)
        res <<= b_vec - at_x
        return norm_2(res)

    def init(self, x_vec, b_vec, any_op, pre_op):  # :106-122
        self._work = [_like(x_vec) for _ in range(4)]  # s, t, r, w
        self.inner_iterations = 0
        return self._residual(x_vec, b_vec, any_op)

    def iterate(self, x_vec, b_vec, any_op, pre_op):  # :124-161
        step, res = self._work[1], self._work[2]
        inner = BiCgStabSolver()
        inner.absolute_error_tolerance = inner.relative_error_tolerance = 1.0e-8
        jacobian = self._linearised_at(x_vec, any_op)
        step <<= res
        inner.solve(step, res, jacobian)
        self.inner_iterations += inner.iteration
        x_vec += step
        return self._residual(x_vec, b_vec, any_op)


def _like(v: DeviceVector) -> DeviceVector:
    w = DeviceVector()
    w.assign(v, False)
    return w


def solve(solver_cls, x_vec: DeviceVector, b_vec: DeviceVector, any_op: Operator) -> bool:
    """``solve<Solver>(x, b, op)``  Solver.hpp:261-265."""
    return solver_cls().solve(x_vec, b_vec, any_op)


def solve_non_uniform(solver: Solver, x_vec: DeviceVector, b_vec: DeviceVector, any_op: Operator) -> bool:
    """Solver.hpp:271-292: ``A(x) = b`` for an operator with ``A(0) != 0`` -- solve ``A(x) - A(0) = b - A(0)``."""
    at_zero, rhs = _like(x_vec), _like(b_vec)
    fill_with(rhs, 0.0)
    any_op.mul(at_zero, rhs)
    rhs <<= b_vec - at_zero

    def uniform(y_vec, in_vec):
        any_op.mul(y_vec, in_vec)
        y_vec -= at_zero

    return solver.solve(x_vec, rhs, make_operator(uniform))
