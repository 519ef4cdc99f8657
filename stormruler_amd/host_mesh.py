"""The library's host-side meshes (``storm_hip_mesh_*``, csrc/mesh_host.hip) from Python: the native Triangle / TetGen
reader (both branches of ``read_mesh_from_tetgen``, source/Storm/Mallard/IoTetgen.hpp:44-235), the cell permutation
hook (MeshUnstructured.hpp:443-459), and the row partition with its halo plan.  The numpy modules
``io_tetgen`` / ``partition`` restate the same rules and are what the tests check this against."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, lib
from .mesh import FaceGraph
from .partition import HaloPlan

__all__ = ["HostMesh", "partition_rcb", "partition_slabs", "write_tetgen"]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None and a.size else None


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


class HostMesh:
    """Owns a ``storm_hip_mesh``."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def read_tetgen(cls, prefix: str, dim: int = 0) -> "HostMesh":
        h = C.c_void_p()
        check(lib.storm_hip_mesh_read_tetgen(prefix.encode(), dim, C.byref(h)))
        return cls(h)

    @classmethod
    def from_simplices(cls, pos, listed, listed_label, cells) -> "HostMesh":
        pos = np.ascontiguousarray(pos, np.float64)
        listed = np.ascontiguousarray(listed, np.int64)
        cells = np.ascontiguousarray(cells, np.int64)
        lab = None if listed_label is None else np.ascontiguousarray(listed_label, np.int64)
        h = C.c_void_p()
        check(lib.storm_hip_mesh_from_simplices(pos.shape[1], pos.shape[0], _p(pos, C.c_double), listed.shape[0],
                                                _p(listed, C.c_int64), _p(lab, C.c_int64), cells.shape[0],
                                                _p(cells, C.c_int64), C.byref(h)))
        return cls(h)

    @classmethod
    def from_face_graph(cls, g: FaceGraph) -> "HostMesh":
        a = lambda v, t: np.ascontiguousarray(v, t)  # noqa: E731
        inner, outer, area = a(g.inner, np.int64), a(g.outer, np.int64), a(g.area, np.float64)
        center, volume = a(g.center, np.float64), a(g.volume, np.float64)
        b_cell, b_area, b_center = a(g.b_cell, np.int64), a(g.b_area, np.float64), a(g.b_center, np.float64)
        gid = None if g.global_id is None else a(g.global_id, np.int64)
        own = None if g.halo_owner is None else a(g.halo_owner, np.int32)
        h = C.c_void_p()
        check(lib.storm_hip_mesh_create(g.dim, g.n_cells, g.n_halo, g.n_faces, _p(inner, C.c_int64), _p(outer, C.c_int64),
                                        _p(area, C.c_double), _p(center, C.c_double), _p(volume, C.c_double), g.n_bfaces,
                                        _p(b_cell, C.c_int64), _p(b_area, C.c_double), _p(b_center, C.c_double),
                                        _p(gid, C.c_int64), _p(own, C.c_int32), C.byref(h)))
        return cls(h)

    def view(self) -> _lib.MeshView:
        v = _lib.MeshView()
        check(lib.storm_hip_mesh_get_view(self._h, C.byref(v)))
        return v

    def face_graph(self) -> FaceGraph:
        """A copy of the mesh's arrays as a :class:`FaceGraph`."""
        v = self.view()
        nt, d = v.n_cells + v.n_halo, v.dim
        return FaceGraph(n_cells=v.n_cells, dim=d, inner=_arr(v.inner, v.n_faces, np.int64), outer=_arr(v.outer, v.n_faces, np.int64),
                         area=_arr(v.area, v.n_faces, np.float64), center=_arr(v.center, nt * d, np.float64).reshape(nt, d),
                         volume=_arr(v.volume, nt, np.float64), b_cell=_arr(v.b_cell, v.n_bfaces, np.int64),
                         b_area=_arr(v.b_area, v.n_bfaces, np.float64),
                         b_center=_arr(v.b_center, v.n_bfaces * d, np.float64).reshape(v.n_bfaces, d), n_halo=v.n_halo,
                         global_id=_arr(v.global_id, nt, np.int64) if v.global_id else None,
                         halo_owner=_arr(v.halo_owner, v.n_halo, np.int32) if v.halo_owner else None)

    def halo_plan(self) -> HaloPlan:
        v = self.view()
        q = v.n_nbrs
        send_ptr = _arr(v.send_ptr, q + 1, np.int64)
        return HaloPlan(_arr(v.nbr_rank, q, np.int32), send_ptr, _arr(v.send_idx, int(send_ptr[-1]), np.int64),
                        _arr(v.recv_ptr, q + 1, np.int64))

    def permute_cells(self, order: np.ndarray) -> None:
        order = np.ascontiguousarray(order, np.int64)
        check(lib.storm_hip_mesh_permute_cells(self._h, _p(order, C.c_int64)))

    def order_cells(self, mode: str = "auto") -> str:
        """Renumber the owned cells by the library's ordering from the cell centres (``storm_hip_order_cells``)."""
        v = self.view()
        order = np.empty(v.n_cells, np.int64)
        kind = C.c_int32(0)
        check(lib.storm_hip_order_cells(v.dim, v.n_cells, v.center, {"auto": 0, "morton": 1, "lattice": 2, "hilbert": 3}[mode],
                                        _p(order, C.c_int64), C.byref(kind)))
        self.permute_cells(order)
        return {1: "lattice", 2: "morton", 3: "hilbert"}.get(kind.value, "none")

    def partition(self, part: np.ndarray, n_parts: int, rank: int) -> "HostMesh":
        part = np.ascontiguousarray(part, np.int32)
        h = C.c_void_p()
        check(lib.storm_hip_mesh_partition(self._h, _p(part, C.c_int32), n_parts, rank, C.byref(h)))
        return HostMesh(h)

    def compute_halo_plan(self, rank: int) -> HaloPlan:
        check(lib.storm_hip_mesh_halo_plan(self._h, rank))
        return self.halo_plan()

    def create_operator(self, ctx):
        """``storm_hip_op_create_from_mesh_object``: the operator (and its halo plan) straight from this mesh."""
        from .api import StencilMatrix

        h = C.c_void_p()
        check(lib.storm_hip_op_create_from_mesh_object(ctx._h, self._h, C.byref(h)))
        return StencilMatrix(ctx, h)

    def close(self):
        if getattr(self, "_h", None):
            lib.storm_hip_mesh_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def partition_rcb(center: np.ndarray, n_parts: int) -> np.ndarray:
    center = np.ascontiguousarray(center, np.float64)
    part = np.empty(center.shape[0], np.int32)
    check(lib.storm_hip_partition_rcb(center.shape[1], center.shape[0], _p(center, C.c_double), n_parts, _p(part, C.c_int32)))
    return part


def partition_slabs(center: np.ndarray, axis: int, n_parts: int) -> np.ndarray:
    center = np.ascontiguousarray(center, np.float64)
    part = np.empty(center.shape[0], np.int32)
    check(lib.storm_hip_partition_slabs(center.shape[1], center.shape[0], _p(center, C.c_double), axis, n_parts,
                                        _p(part, C.c_int32)))
    return part


def write_tetgen(prefix: str, pos, listed, listed_label: Optional[np.ndarray], cells) -> None:
    pos = np.ascontiguousarray(pos, np.float64)
    listed = np.ascontiguousarray(listed, np.int64)
    cells = np.ascontiguousarray(cells, np.int64)
    lab = None if listed_label is None else np.ascontiguousarray(listed_label, np.int64)
    check(lib.storm_hip_mesh_write_tetgen(prefix.encode(), pos.shape[1], pos.shape[0], _p(pos, C.c_double), listed.shape[0],
                                          _p(listed, C.c_int64), _p(lab, C.c_int64), cells.shape[0], _p(cells, C.c_int64)))
