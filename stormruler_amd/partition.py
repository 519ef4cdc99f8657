"""Row (cell) partition of a face graph across ranks, with halo plans.

Host-side numpy only.  A rank's local graph numbers its owned cells first (ascending global id
unless an ordering is applied afterwards) and appends the halo cells grouped by owner rank, each
group in ascending global id.  The send list towards a neighbour is the set of owned cells that
share a face with one of that neighbour's cells, again in ascending global id -- which is exactly
the neighbour's halo group for this rank, so no index lists ever have to be exchanged.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

from .mesh import FaceGraph, structured_box_slab

__all__ = ["HaloPlan", "partition_graph", "halo_plan", "slab_partition", "slab_ranges", "rcb_partition"]


@dataclass
class HaloPlan:
    nbr_rank: np.ndarray  # int32 [Q]
    send_ptr: np.ndarray  # int64 [Q + 1]
    send_idx: np.ndarray  # int64 [send_ptr[-1]] owned local rows to pack
    recv_ptr: np.ndarray  # int64 [Q + 1] offsets into the halo tail

    @property
    def n_nbrs(self) -> int:
        return int(self.nbr_rank.size)


def halo_plan(g: FaceGraph, rank: int) -> HaloPlan:
    """Halo plan of a local graph whose ``global_id`` / ``halo_owner`` are set."""
    n = g.n_cells
    if g.n_halo == 0:
        z = np.zeros(1, np.int64)
        return HaloPlan(np.zeros(0, np.int32), z, np.zeros(0, np.int64), z.copy())
    owner = np.asarray(g.halo_owner)
    assert np.all(owner >= 0) and np.all(owner != rank)
    assert np.all(np.diff(owner) >= 0), "halo cells must be grouped by ascending owner rank"
    nbrs = np.unique(owner)
    recv_ptr = np.concatenate([[0], np.cumsum([(owner == q).sum() for q in nbrs])]).astype(np.int64)
    for a, b in zip(recv_ptr[:-1], recv_ptr[1:]):
        assert np.all(np.diff(g.global_id[n + a:n + b]) > 0), "halo group not in ascending global id"
    # owned cells adjacent to a halo cell, per owner of that halo cell
    in_own, out_own = g.inner < n, g.outer < n
    m1 = in_own & ~out_own   # inner owned, outer halo
    m2 = out_own & ~in_own   # outer owned, inner halo
    own_cell = np.concatenate([g.inner[m1], g.outer[m2]])
    hal_cell = np.concatenate([g.outer[m1], g.inner[m2]]) - n
    send_idx: List[np.ndarray] = []
    send_ptr = [0]
    for q in nbrs:
        cells = np.unique(own_cell[owner[hal_cell] == q])
        cells = cells[np.argsort(g.global_id[cells], kind="stable")]
        send_idx.append(cells)
        send_ptr.append(send_ptr[-1] + cells.size)
    return HaloPlan(nbrs.astype(np.int32), np.asarray(send_ptr, np.int64),
                    np.concatenate(send_idx).astype(np.int64) if send_idx else np.zeros(0, np.int64), recv_ptr)


def partition_graph(g: FaceGraph, part: np.ndarray, rank: int) -> FaceGraph:
    """Local graph of ``rank`` for the cell -> rank map ``part`` of a single-rank graph ``g``."""
    assert g.n_halo == 0 and part.shape == (g.n_cells,)
    owned = np.flatnonzero(part == rank).astype(np.int64)
    is_own = part == rank
    touch = is_own[g.inner] | is_own[g.outer]
    fi, fo = g.inner[touch], g.outer[touch]
    other = np.concatenate([fi[~is_own[fi]], fo[~is_own[fo]]])
    halo = np.unique(other)
    halo = halo[np.lexsort((halo, part[halo]))]  # by owner, then global id
    gid = np.concatenate([owned, halo])
    loc = np.full(g.n_cells, -1, np.int64)
    loc[gid] = np.arange(gid.size)
    bsel = is_own[g.b_cell] if g.n_bfaces else np.zeros(0, bool)
    return FaceGraph(n_cells=owned.size, dim=g.dim, inner=loc[fi], outer=loc[fo], area=g.area[touch],
                     center=g.center[gid], volume=g.volume[gid], b_cell=loc[g.b_cell[bsel]],
                     b_area=g.b_area[bsel], b_center=g.b_center[bsel], n_halo=halo.size, global_id=gid,
                     halo_owner=part[halo].astype(np.int32))


def slab_ranges(nz_glob: int, n_ranks: int):
    """Contiguous z-ranges, as even as possible."""
    base, rem = divmod(nz_glob, n_ranks)
    out, k = [], 0
    for r in range(n_ranks):
        k1 = k + base + (1 if r < rem else 0)
        out.append((k, k1))
        k = k1
    return out


def slab_partition(nx: int, ny: int, nz_per_rank: int, n_ranks: int, rank: int, dirichlet: bool = True):
    """The weak-scaling layout of BASELINE configs 2/3: an nx*ny*(nz_per_rank*n_ranks) box in a
    unit-spacing-preserving domain (lengths scale with the rank count so h stays 1/nx), rank r
    owning z-planes [r*nz_per_rank, (r+1)*nz_per_rank).  Returns (local graph, halo plan)."""
    nz_glob = nz_per_rank * n_ranks
    k0, k1 = rank * nz_per_rank, (rank + 1) * nz_per_rank
    lengths = (1.0, ny / nx, nz_glob / nx)
    g = structured_box_slab(nx, ny, nz_glob, k0, k1, lengths, dirichlet,
                            rank_of_k=lambda k: k // nz_per_rank)
    return g, halo_plan(g, rank)


def rcb_partition(center: np.ndarray, n_parts: int) -> np.ndarray:
    """Recursive coordinate bisection of the cell centres into ``n_parts`` balanced parts.

    The build's own k-way partitioner for general meshes (METIS is not available): split
    the longest axis of the bounding box at the k-th smallest (coordinate, cell id), recurse
    (``storm_hip_partition_rcb`` is the same rule natively: csrc/mesh_host.hip).  Part sizes differ by at most
    one cell per level; works for any ``n_parts`` (the split is proportional, not only powers of two).
    Returns the cell -> rank map for :func:`partition_graph`.
    """
    n = center.shape[0]
    part = np.zeros(n, np.int64)

    def split(idx: np.ndarray, first: int, count: int) -> None:
        if count == 1:
            part[idx] = first
            return
        left_parts = count // 2
        pts = center[idx]
        axis = int(np.argmax(pts.max(axis=0) - pts.min(axis=0)))
        k = (idx.size * left_parts) // count
        order = np.lexsort((idx, pts[:, axis]))  # ties by cell id: the cut does not depend on the recursion's history
        split(idx[order[:k]], first, left_parts)
        split(idx[order[k:]], first + left_parts, count - left_parts)

    split(np.arange(n, dtype=np.int64), 0, int(n_parts))
    return part
