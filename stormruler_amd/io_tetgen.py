"""Triangle / TetGen mesh ingestion -> face graph (2-D and 3-D).  Host preprocessing, numpy only.

Follows ``read_mesh_from_tetgen`` (source/Storm/Mallard/IoTetgen.hpp:44-235), both of its branches
(``mesh3D = mesh_dim_v<Mesh> == 3``, :47):

* ``.node``: header ``n dim n_attr has_marker``, then ``id x y [z] [attrs] [marker]`` (:55-100; z at :87);
* ``.edge``: header ``n has_marker``, then ``id n1 n2 [marker]`` (:103-137) -- read in both dimensions; in 3-D
  the edges do not enter the face graph (the file must exist and parse, as the reference demands);
* ``.face`` (3-D only): header ``n has_marker``, then ``id n1 n2 n3 [marker]`` (:139-175);
* ``.ele`` : header ``n nodes_per_cell has_attr``, then ``id n1 n2 n3 [n4] [attr]`` (:177-217; n4 at :207-209);
* ``#`` starts a comment to the end of the line (``FilteringStreambuf<'#','\\n'>``, :61);
* node ids are used exactly as written (the reference does not rebase them: zero-based files, ``-z``);
* "TetGen may not generate all the edges/faces" (:219-221): a cell side that the ``.edge`` / ``.face`` file does
  not list is created when the first cell that owns it is inserted (``find_or_insert``,
  MeshUnstructured.hpp:431-437, called for the cell's parts in the order of ``Triangle::edges()``
  Shape.hpp:303-305 / ``Tetrahedron::faces()`` :590-594) and gets label 0; listed sides keep their file order and
  their marker as label.  ``assign_labels`` (MeshUnstructured.hpp:464-500) stable-sorts the sides by label, so
  the interior sides (label 0, ``interior_faces()``) are: the listed ones with marker 0 in file order, then the
  created ones in order of creation;
* a side's inner cell is the first inserted cell that owns it, the outer cell the second
  (``_update_face_orientation``, MeshUnstructured.hpp:509-554); the second cell must see the side with the
  opposite orientation (``STORM_ENSURE``, :546-548) -- checked here, a ``RuntimeError`` otherwise;
* geometry: cell centre = mean of the nodes, summed left to right (Shape.hpp:155-167; tetrahedron :598-607);
  2-D cell "volume" ``0.5 |d0.x d1.y - d0.y d1.x|`` (:309-321), side "area" = edge length (:242-247);
  3-D side area ``length(cross(v2 - v1, v3 - v1)) / 2`` (:321) on the nodes as inserted.  **The reference has no
  ``volume(Tetrahedron)``** (only ``barycenter``, SURVEY.md headline fact 4): the tetrahedron volume
  ``|det[v2 - v1, v3 - v1, v4 - v1]| / 6`` (cofactor expansion along the first column, left to right) is this
  build's own completion of the 3-D branch, the analogue of the 2-D formula.

``read_tetgen`` is the restatement the native reader of the library (``storm_hip_mesh_read_tetgen``,
csrc/mesh_host.hip) is checked against, array for array (tests/test_tetgen_mesh.py).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import numpy as np

from .mesh import FaceGraph

__all__ = ["read_tetgen", "read_triangle", "face_graph_from_simplices", "write_tetgen", "tet_box"]


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


def _rows(body, n: int, stride: int, dtype, what: str, path: str) -> np.ndarray:
    if len(body) < n * stride:
        raise RuntimeError(f"Cannot read the {what} from file '{path}'!")
    return np.array(body[: n * stride], dtype=dtype).reshape(n, stride)


def _pack_keys(nodes_sorted: np.ndarray, n_nodes: int) -> np.ndarray:
    """One int64 key per row of ascending node ids.  Two ids pack directly; a third is packed onto the RANK of the
    first pair (n_nodes^3 overflows 63 bits beyond 2 M nodes)."""
    key = nodes_sorted[:, 0] * np.int64(n_nodes) + nodes_sorted[:, 1]
    if nodes_sorted.shape[1] == 3:
        _, rank = np.unique(key, return_inverse=True)
        key = rank.astype(np.int64) * np.int64(n_nodes) + nodes_sorted[:, 2]
    return key


def _parity(nodes: np.ndarray) -> np.ndarray:
    """+1 / -1: the parity of the permutation that sorts each row (2 or 3 distinct ids)."""
    if nodes.shape[1] == 2:
        return np.where(nodes[:, 0] < nodes[:, 1], 1, -1).astype(np.int8)
    a, b, c = nodes[:, 0], nodes[:, 1], nodes[:, 2]
    inv = (a > b).astype(np.int8) + (a > c) + (b > c)
    return np.where(inv % 2 == 0, 1, -1).astype(np.int8)


def face_graph_from_simplices(pos: np.ndarray, listed: np.ndarray, listed_label: Optional[np.ndarray],
                              cells: np.ndarray) -> FaceGraph:
    """The face graph of a simplicial mesh as the reference's ``insert`` calls build it: ``pos [n_nodes, dim]``,
    ``listed [n_listed, dim]`` the sides of the ``.edge`` (2-D) / ``.face`` (3-D) file with their markers (``None``:
    all 0), ``cells [n_cells, dim + 1]``.  See the module docstring for the rules."""
    n_nodes, dim = pos.shape
    n_cells = cells.shape[0]
    assert dim in (2, 3) and cells.shape[1] == dim + 1 and listed.shape[1] == dim
    n_listed = listed.shape[0]
    lo = min(int(cells.min()) if n_cells else 0, int(listed.min()) if n_listed else 0)
    hi = max(int(cells.max()) if n_cells else 0, int(listed.max()) if n_listed else 0)
    if lo < 0 or hi >= n_nodes:
        raise RuntimeError("node index out of range (files must be zero-based, `triangle -z` / `tetgen -z`)")
    # the sides of a cell in the order its insertion visits them
    if dim == 2:
        part = [(0, 1), (1, 2), (2, 0)]  # Triangle::edges(), Shape.hpp:303-305
    else:
        part = [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)]  # Tetrahedron::faces(), Shape.hpp:590-594
    cf = np.stack([cells[:, list(p)] for p in part], axis=1).reshape(n_cells * (dim + 1), dim)
    cell_of = np.repeat(np.arange(n_cells, dtype=np.int64), dim + 1)
    both = np.concatenate([listed, cf]) if n_listed else cf
    key = _pack_keys(np.sort(both, axis=1), n_nodes)
    uniq, first_at, inv = np.unique(key, return_index=True, return_inverse=True)
    if n_listed and np.unique(key[:n_listed]).size != n_listed:
        raise RuntimeError("a side is listed twice")
    # side ids in order of first appearance: the listed ones in file order, then by creation
    by_first = np.argsort(first_at, kind="stable")
    side_id = np.empty(uniq.size, np.int64)
    side_id[by_first] = np.arange(uniq.size, dtype=np.int64)
    n_sides = uniq.size
    side_of_cf = side_id[inv[n_listed:]]
    n_own = np.bincount(side_of_cf, minlength=n_sides)
    if np.any(n_own > 2) or np.any(n_own == 0):
        raise RuntimeError("Invalid number of the face cells!")  # STORM_ABORT, MeshUnstructured.hpp:550
    first = np.full(n_sides, -1, np.int64)
    second = np.full(n_sides, -1, np.int64)
    # cell_of ascends: a reversed assignment leaves the first owner in `first`, a forward one the last in `second`
    second[side_of_cf] = cell_of
    first[side_of_cf[::-1]] = cell_of[::-1]
    # the second owner must see the side reversed (MeshUnstructured.hpp:546-548)
    par = _parity(cf)
    psum = np.zeros(n_sides, np.int64)
    np.add.at(psum, side_of_cf, par)
    if np.any((n_own == 2) & (psum != 0)):
        raise RuntimeError("Face has two adjacent cells, but the second cell cannot be the outer one!")
    label = np.zeros(n_sides, np.int64)
    if n_listed and listed_label is not None:
        label[:n_listed] = listed_label
    if np.any((label == 0) & (n_own != 2)):
        raise RuntimeError("an unlabelled side has a single adjacent cell")
    # the nodes a side was inserted with: the file's for listed sides, the creating cell's for the others
    side_nodes = np.empty((n_sides, dim), np.int64)
    side_nodes[side_id[inv[::-1]]] = both[::-1]  # (repeated indices: the last assignment stands = the FIRST appearance)
    # geometry
    p = [pos[cells[:, k]] for k in range(dim + 1)]
    if dim == 2:
        center = ((p[0] + p[1]) + p[2]) / 3.0
        d0, d1 = p[1] - p[0], p[2] - p[0]
        volume = 0.5 * np.abs(d0[:, 0] * d1[:, 1] - d0[:, 1] * d1[:, 0])
        ev = pos[side_nodes[:, 1]] - pos[side_nodes[:, 0]]
        area = np.sqrt(0.0 + ev[:, 0] * ev[:, 0] + ev[:, 1] * ev[:, 1])
        mid = 0.5 * (pos[side_nodes[:, 0]] + pos[side_nodes[:, 1]])
    else:
        center = (((p[0] + p[1]) + p[2]) + p[3]) / 4.0  # Shape.hpp:601-606
        a, b, c = p[1] - p[0], p[2] - p[0], p[3] - p[0]
        det = (a[:, 0] * (b[:, 1] * c[:, 2] - b[:, 2] * c[:, 1])
               - a[:, 1] * (b[:, 0] * c[:, 2] - b[:, 2] * c[:, 0])
               + a[:, 2] * (b[:, 0] * c[:, 1] - b[:, 1] * c[:, 0]))
        volume = np.abs(det) / 6.0
        q1, q2, q3 = pos[side_nodes[:, 0]], pos[side_nodes[:, 1]], pos[side_nodes[:, 2]]
        u, v = q2 - q1, q3 - q1
        cx = u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1]
        cy = u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2]
        cz = u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]
        area = np.sqrt(((0.0 + cx * cx) + cy * cy) + cz * cz) / 2.0  # Shape.hpp:321
        mid = ((q1 + q2) + q3) / 3.0
    interior = label == 0
    boundary = ~interior
    g = FaceGraph(n_cells=n_cells, dim=dim, inner=first[interior], outer=second[interior], area=area[interior],
                  center=center, volume=volume, b_cell=first[boundary], b_area=area[boundary], b_center=mid[boundary])
    g.validate()
    return g


def _read_arrays(prefix: str, dim: Optional[int]) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray], np.ndarray]:
    prefix = prefix[:-1] if prefix.endswith(".") else prefix
    path = prefix + ".node"
    hdr, body = _table(path, 4)
    n_nodes, fdim, n_attr, has_marker = int(hdr[0]), int(hdr[1]), int(hdr[2]), int(hdr[3])
    if fdim not in (2, 3) or (dim is not None and fdim != dim):
        raise RuntimeError(f"Unexpected number of the dimensions in node file '{path}' header! "
                           f"Expected {dim if dim is not None else '2 or 3'}, got {fdim}.")
    dim = fdim
    a = _rows(body, n_nodes, 1 + dim + n_attr + (1 if has_marker else 0), np.float64, "nodes", path)
    pos = np.ascontiguousarray(a[:, 1:1 + dim])

    path = prefix + ".edge"
    hdr, body = _table(path, 2)
    n_edges, e_marker = int(hdr[0]), int(hdr[1])
    e = _rows(body, n_edges, 3 + (1 if e_marker else 0), np.int64, "edges", path)
    if n_edges and (e[:, 1:3].min() < 0 or e[:, 1:3].max() >= n_nodes):
        raise RuntimeError("node index out of range (files must be zero-based)")
    if dim == 2:
        listed, label = e[:, 1:3], (e[:, 3] if e_marker else None)
    else:
        path = prefix + ".face"
        hdr, body = _table(path, 2)
        n_faces, f_marker = int(hdr[0]), int(hdr[1])
        f = _rows(body, n_faces, 4 + (1 if f_marker else 0), np.int64, "faces", path)
        listed, label = f[:, 1:4], (f[:, 4] if f_marker else None)

    path = prefix + ".ele"
    hdr, body = _table(path, 3)
    n_cells, npc, c_attr = int(hdr[0]), int(hdr[1]), int(hdr[2])
    if npc != dim + 1:
        raise RuntimeError(f"Unexpected number of the nodes per cell in the cell file '{path}' header! "
                           f"Expected {dim + 1}, got {npc}.")
    c = _rows(body, n_cells, 1 + npc + (1 if c_attr else 0), np.int64, "cells", path)
    return pos, np.ascontiguousarray(listed), label, np.ascontiguousarray(c[:, 1:1 + npc])


def read_tetgen(prefix: str, dim: Optional[int] = None) -> FaceGraph:
    """Read ``<prefix>node/.edge/[.face]/.ele`` (``prefix`` ends with ``.1.`` or ``.1``) into a face graph; ``dim`` =
    the mesh dimension the caller expects (``mesh_dim_v<Mesh>``; a mismatch with the node file's header is the
    reference's I/O error), ``None`` = whatever the node file says.

    Sides with a marker != 0 become ``b_cell/b_area/b_center`` with the side's barycentre as face centre; the
    reference's stencil ignores them (pure Neumann, Playground.cpp:119)."""
    return face_graph_from_simplices(*_read_arrays(prefix, dim))


def read_triangle(prefix: str) -> FaceGraph:
    """The 2-D case (``Feathers::Mesh`` is ``UnstructuredMesh<2, 2, CsrTable>``, Feathers/Field.hpp:51)."""
    return read_tetgen(prefix, 2)


# ---------------------------------------------------------------------------------------------------------
# A seeded tetrahedral test mesh and a writer of the file format (test / bench infrastructure: the reference has
# neither; its 3-D inputs would come from TetGen itself).


def tet_box(n: int, jitter: float = 0.15, seed: int = 7, lengths=(1.0, 1.0, 1.0)):
    """The n^3 box cut into 6 tetrahedra per cube (Kuhn's subdivision: the six monotone paths from a cube's
    corner (0,0,0) to (1,1,1); every cube cut alike, so neighbouring cubes agree on their shared squares' diagonals
    and the mesh conforms), all positively oriented, interior nodes displaced by at most ``jitter`` of the spacing
    per coordinate (seeded; below 1/6 no cell can invert: its edge matrix stays diagonally dominant): 6 n^3 cells of six different shapes and all-distinct face weights, rows of 4 neighbours
    (2 - 3 at the walls).  Returns ``(pos [(n+1)^3, 3], boundary faces [12 n^2, 3], cells [6 n^3, 4])`` -- what
    ``tetgen`` itself writes: ``.face`` lists the boundary triangles only (marker 1)."""
    m = n + 1
    idx = np.arange(m, dtype=np.int64)
    kk, jj, ii = np.meshgrid(idx, idx, idx, indexing="ij")  # node id = (k m + j) m + i
    pos = np.stack([ii.ravel() * (lengths[0] / n), jj.ravel() * (lengths[1] / n), kk.ravel() * (lengths[2] / n)], axis=1)
    if jitter > 0:
        rng = np.random.default_rng(seed)
        d = (2.0 * rng.random(pos.shape) - 1.0) * (jitter * np.array(lengths) / n)
        inside = ((ii > 0) & (ii < n) & (jj > 0) & (jj < n) & (kk > 0) & (kk < n)).ravel()
        pos[inside] += d[inside]
    c = np.arange(n, dtype=np.int64)
    ck, cj, ci = np.meshgrid(c, c, c, indexing="ij")
    base = ((ck * m + cj) * m + ci).ravel()  # the cube's corner (0,0,0); cubes in lexicographic order
    step = np.array([1, m, m * m], dtype=np.int64)
    tets = []
    for perm in ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)):
        v0 = base
        v1 = v0 + step[perm[0]]
        v2 = v1 + step[perm[1]]
        v3 = v2 + step[perm[2]]
        # the path's orientation is the permutation's sign: swap two nodes of the odd ones so that every det > 0
        odd = perm in ((0, 2, 1), (1, 0, 2), (2, 1, 0))
        tets.append(np.stack([v0, v2, v1, v3] if odd else [v0, v1, v2, v3], axis=1))
    cells = np.stack(tets, axis=1).reshape(-1, 4)  # cube-major, six cells each
    # boundary triangles: the sides whose three nodes lie in one wall of the box (only cells of the outer cube layer can
    # have one), listed in cell order with the owner's (outward) orientation
    wall_cube = ((ci == 0) | (ci == n - 1) | (cj == 0) | (cj == n - 1) | (ck == 0) | (ck == n - 1)).ravel()
    cand = cells[np.repeat(wall_cube, 6)]
    part = [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)]
    cf = np.stack([cand[:, list(p)] for p in part], axis=1).reshape(-1, 3)
    on_wall = np.zeros(cf.shape[0], bool)
    for coord in (cf % m, (cf // m) % m, cf // (m * m)):
        for w in (0, n):
            on_wall |= np.all(coord == w, axis=1)
    bfaces = cf[on_wall]
    return pos, bfaces, cells


def write_tetgen(prefix: str, pos: np.ndarray, listed: np.ndarray, listed_label: Optional[np.ndarray], cells: np.ndarray,
                 comment: str = "") -> None:
    """Write ``<prefix>.node / .edge / [.face] / .ele`` in the format ``read_mesh_from_tetgen`` reads (zero-based ids,
    ``%.17g`` coordinates: the doubles round-trip exactly).  3-D: ``.edge`` is written with zero entries (TetGen "may
    not generate all the edges", IoTetgen.hpp:219-221)."""
    prefix = prefix[:-1] if prefix.endswith(".") else prefix
    n_nodes, dim = pos.shape
    head = f"# {comment}\n" if comment else ""

    def table(path, header, ids, cols, fmt):
        with open(path, "w") as f:
            f.write(head + header + "\n")
            if ids.size:
                np.savetxt(f, np.column_stack([ids] + cols), fmt=fmt)

    with open(prefix + ".node", "w") as f:
        f.write(head + f"{n_nodes} {dim} 0 0\n")
        for i in range(0, n_nodes, 1 << 18):  # (np.savetxt of mixed int / float columns: row by row, in slabs)
            blk = pos[i:i + (1 << 18)]
            cols = [np.char.mod("%d", np.arange(i, i + blk.shape[0]))] + [np.char.mod("%.17g", blk[:, k]) for k in range(dim)]
            rows = cols[0]
            for c_ in cols[1:]:
                rows = np.char.add(np.char.add(rows, " "), c_)
            f.write("\n".join(rows.tolist()) + "\n")
    lab = listed_label if listed_label is not None else np.zeros(listed.shape[0], np.int64)
    ids = np.arange(listed.shape[0], dtype=np.int64)
    if dim == 2:
        table(prefix + ".edge", f"{listed.shape[0]} 1", ids, [listed, lab], "%d")
    else:
        with open(prefix + ".edge", "w") as f:
            f.write(head + "0 1\n")
        table(prefix + ".face", f"{listed.shape[0]} 1", ids, [listed, lab], "%d")
    table(prefix + ".ele", f"{cells.shape[0]} {dim + 1} 0", np.arange(cells.shape[0], dtype=np.int64), [cells], "%d")
