"""Triangle (2-D TetGen-format) mesh ingestion -> face graph.  Host preprocessing only.

Follows ``read_mesh_from_tetgen`` (source/Storm/Mallard/IoTetgen.hpp:44-235) for the 2-D case the
reference supports:

* ``.node``: header ``n dim n_attr has_marker``, then ``id x y [attrs] [marker]`` (:55-100);
* ``.edge``: header ``n has_marker``, then ``id n1 n2 [marker]`` (:103-137);
* ``.ele`` : header ``n nodes_per_cell has_attr``, then ``id n1 n2 n3 [attr]`` (:177-217);
* ``#`` starts a comment to the end of the line (``FilteringStreambuf<'#','\\n'>``, :61);
* node ids are used exactly as written (the reference does not rebase them; its files are made
  with ``triangle -z``, i.e. zero-based);
* edge markers become labels; ``assign_labels`` (MeshUnstructured.hpp:464-500) stable-sorts the
  edges by label, so the interior faces (label 0, ``interior_faces()``) keep their file order;
* a face's inner cell is the first inserted cell that owns it, the outer cell the second
  (MeshUnstructured.hpp:509-554): with cells inserted in ``.ele`` order, inner = lower cell id;
* geometry: cell centre = mean of the three nodes (Shape.hpp:155-167), cell "volume" =
  ``0.5 * |d0.x d1.y - d0.y d1.x|`` (Shape.hpp:309-321), face "area" = edge length (:242-247).
"""
from __future__ import annotations

import os
from typing import List

import numpy as np

from .mesh import FaceGraph

__all__ = ["read_triangle"]


def _tokens(path: str) -> List[str]:
    opener = open
    if not os.path.exists(path) and os.path.exists(path + ".gz"):  # fixtures are stored compressed
        import gzip

        path, opener = path + ".gz", lambda p: gzip.open(p, "rt")
    if not os.path.exists(path):
        raise RuntimeError(f"Cannot open the file '{path}'!")  # STORM_THROW_IO -> std::runtime_error
    out: List[str] = []
    with opener(path) as f:
        for line in f:
            out.extend(line.split("#", 1)[0].split())
    return out


def _table(path: str, header_len: int):
    t = _tokens(path)
    if len(t) < header_len:
        raise RuntimeError(f"Cannot read the file '{path}' header!")
    return t[:header_len], t[header_len:]


def read_triangle(prefix: str) -> FaceGraph:
    """Read ``<prefix>node/.edge/.ele`` (``prefix`` ends with ``.1.`` or ``.1``) into a 2-D face graph.

    Boundary edges (marker != 0) become ``b_cell/b_area/b_center`` with the edge midpoint as face
    centre; the reference's stencil ignores them (pure Neumann, Playground.cpp:119).
    """
    prefix = prefix[:-1] if prefix.endswith(".") else prefix
    hdr, body = _table(prefix + ".node", 4)
    n_nodes, dim, n_attr, has_marker = int(hdr[0]), int(hdr[1]), int(hdr[2]), int(hdr[3])
    if dim != 2:
        raise RuntimeError(f"Unexpected number of the dimensions in node file: expected 2, got {dim}.")
    stride = 1 + dim + n_attr + (1 if has_marker else 0)
    if len(body) < n_nodes * stride:
        raise RuntimeError(f"Cannot read the nodes from file '{prefix}.node'!")
    a = np.array(body[: n_nodes * stride], dtype=np.float64).reshape(n_nodes, stride)
    pos = np.ascontiguousarray(a[:, 1:3])

    hdr, body = _table(prefix + ".edge", 2)
    n_edges, e_marker = int(hdr[0]), int(hdr[1])
    stride = 3 + (1 if e_marker else 0)
    if len(body) < n_edges * stride:
        raise RuntimeError(f"Cannot read the edges from file '{prefix}.edge'!")
    e = np.array(body[: n_edges * stride], dtype=np.int64).reshape(n_edges, stride)
    edge_nodes = e[:, 1:3]
    edge_label = e[:, 3] if e_marker else np.zeros(n_edges, np.int64)

    hdr, body = _table(prefix + ".ele", 3)
    n_cells, npc, c_attr = int(hdr[0]), int(hdr[1]), int(hdr[2])
    if npc != 3:
        raise RuntimeError(f"Unexpected number of the nodes per cell: expected 3, got {npc}.")
    stride = 4 + (1 if c_attr else 0)
    if len(body) < n_cells * stride:
        raise RuntimeError(f"Cannot read the cells from file '{prefix}.ele'!")
    c = np.array(body[: n_cells * stride], dtype=np.int64).reshape(n_cells, stride)
    tri = c[:, 1:4]
    if tri.max() >= n_nodes or edge_nodes.max() >= n_nodes or min(tri.min(), edge_nodes.min()) < 0:
        raise RuntimeError("node index out of range (files must be zero-based, `triangle -z`)")

    # geometry
    p1, p2, p3 = pos[tri[:, 0]], pos[tri[:, 1]], pos[tri[:, 2]]
    center = ((p1 + p2) + p3) / 3.0
    d0, d1 = p2 - p1, p3 - p1
    volume = 0.5 * np.abs(d0[:, 0] * d1[:, 1] - d0[:, 1] * d1[:, 0])
    ev = pos[edge_nodes[:, 1]] - pos[edge_nodes[:, 0]]
    length = np.sqrt(0.0 + ev[:, 0] * ev[:, 0] + ev[:, 1] * ev[:, 1])

    # edge -> cells, cells visited in .ele order (inner = first owner)
    key = lambda a_, b_: np.minimum(a_, b_) * np.int64(n_nodes) + np.maximum(a_, b_)  # noqa: E731
    ekey = key(edge_nodes[:, 0], edge_nodes[:, 1])
    order = np.argsort(ekey, kind="stable")
    sorted_keys = ekey[order]
    ckeys = np.stack([key(tri[:, 0], tri[:, 1]), key(tri[:, 1], tri[:, 2]), key(tri[:, 2], tri[:, 0])], axis=1).ravel()
    cell_of = np.repeat(np.arange(n_cells, dtype=np.int64), 3)
    where = np.searchsorted(sorted_keys, ckeys)
    if np.any(where >= n_edges) or np.any(sorted_keys[np.minimum(where, n_edges - 1)] != ckeys):
        raise RuntimeError("a cell side is missing from the .edge file (run triangle with -e)")
    edge_of = order[where]
    first = np.full(n_edges, -1, np.int64)
    second = np.full(n_edges, -1, np.int64)
    # cell_of is ascending, so a reversed assignment leaves the smallest owner in `first`
    second[edge_of] = cell_of
    first[edge_of[::-1]] = cell_of[::-1]
    n_own = np.bincount(edge_of, minlength=n_edges)
    if np.any(n_own > 2) or np.any(n_own == 0):
        raise RuntimeError("Invalid number of the face cells!")  # STORM_ABORT in the reference
    interior = (edge_label == 0) & (n_own == 2)
    boundary = ~interior
    if np.any((edge_label == 0) & (n_own != 2)):
        raise RuntimeError("an unlabelled edge has a single adjacent cell")
    mid = 0.5 * (pos[edge_nodes[:, 0]] + pos[edge_nodes[:, 1]])
    g = FaceGraph(n_cells=n_cells, dim=2, inner=first[interior], outer=second[interior], area=length[interior],
                  center=center, volume=volume, b_cell=first[boundary], b_area=length[boundary],
                  b_center=mid[boundary])
    g.validate()
    return g
