"""Host-side face graphs: the data contract of the stencil hot path.

The reference's stencil (``stormDivGrad``, source_apps/playground/Playground.cpp:115-131)
reads, per interior face, the two adjacent cells (``FaceView::inner_cell/outer_cell``,
source/Storm/Mallard/Mesh.hpp:269-280), the face area (:254), and per cell the volume and
centre (:304-311).  A :class:`FaceGraph` holds exactly those arrays, plus the Dirichlet
boundary faces the Poisson configs need.

The reference can only build 2-D Triangle meshes (``mesh_dim_v`` is 2,
source/Storm/Mallard/Fwd.hpp:81-85), so the N^3 "structured-as-unstructured" meshes of
BASELINE.json come from :func:`structured_box`, which emits the same kind of face list an
unstructured mesh would: cell id ``(k*ny + j)*nx + i``, faces in cell-major order
+x, +y, +z with inner = lower cell id (the reference's inner/outer convention,
source/Storm/Mallard/MeshUnstructured.hpp:504-554).

Everything here is numpy on the host; nothing in this module touches a device.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np

__all__ = [
    "FaceGraph",
    "structured_box",
    "structured_box_slab",
    "random_permutation",
    "jitter_geometry",
    "tile_ordering",
    "rcm_ordering",
    "geometric_ordering",
    "permute_cells",
    "face_coefficients",
    "convection_diffusion_weights",
    "assemble_csr",
]


@dataclass
class FaceGraph:
    """Face->cell adjacency plus the geometry the stencil reads."""

    n_cells: int
    dim: int
    inner: np.ndarray  # int64 [F]
    outer: np.ndarray  # int64 [F]
    area: np.ndarray  # float64 [F]
    center: np.ndarray  # float64 [n_total_cells, dim]  (owned + halo cells)
    volume: np.ndarray  # float64 [n_total_cells]
    b_cell: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    b_area: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    b_center: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float64))
    # Distributed view (single rank: n_halo == 0, everything below is trivial).
    n_halo: int = 0
    global_id: Optional[np.ndarray] = None  # int64 [n_cells + n_halo]
    halo_owner: Optional[np.ndarray] = None  # int32 [n_halo]

    @property
    def n_faces(self) -> int:
        return int(self.inner.shape[0])

    @property
    def n_bfaces(self) -> int:
        return int(self.b_cell.shape[0])

    @property
    def n_total(self) -> int:
        return self.n_cells + self.n_halo

    def validate(self) -> None:
        nt = self.n_total
        assert self.inner.dtype == np.int64 and self.outer.dtype == np.int64
        assert self.inner.shape == self.outer.shape == self.area.shape
        assert self.center.shape == (nt, self.dim), (self.center.shape, nt, self.dim)
        assert self.volume.shape == (nt,)
        if self.n_faces:
            assert self.inner.min() >= 0 and self.inner.max() < nt
            assert self.outer.min() >= 0 and self.outer.max() < nt
            assert np.all(self.inner != self.outer)
        if self.n_bfaces:
            assert self.b_cell.min() >= 0 and self.b_cell.max() < self.n_cells
        assert np.all(self.volume > 0)


def face_coefficients(g: FaceGraph) -> Tuple[np.ndarray, np.ndarray]:
    """Per-face transmissibility ``A_f / d_f`` for interior and boundary faces.

    ``d_f`` is the centre distance the reference recomputes in every apply
    (``length(cell_outer.center() - cell_inner.center())``, Playground.cpp:126-127); for a
    boundary face it is the cell-centre -> face-centre distance (ghost state at the face).
    Summed in the same left-to-right order as ``length`` (MatrixAlgorithms.hpp:262-270).
    """
    d = g.center[g.outer] - g.center[g.inner]
    s = np.zeros(g.n_faces)
    for k in range(g.dim):
        s = s + d[:, k] * d[:, k]
    coef = g.area / np.sqrt(s)
    if g.n_bfaces:
        db = g.b_center - g.center[g.b_cell]
        sb = np.zeros(g.n_bfaces)
        for k in range(g.dim):
            sb = sb + db[:, k] * db[:, k]
        b_coef = g.b_area / np.sqrt(sb)
    else:
        b_coef = np.zeros(0)
    return coef, b_coef


def convection_diffusion_weights(g: FaceGraph, nu: float, vel: Sequence[float]):
    """Face weights of ``A = -nu L + C(v)`` (BASELINE config 4), for ``StencilMatrix.from_face_weights``.

    ``C(v)`` is the first-order upwind divergence of ``v c`` with constant velocity: the interior /
    boundary face loops of ``UpwindConvectionScheme`` (Feathers/ConvectionScheme.hpp:80-107) with the
    flux ``vn > 0 ? vn c_in : vn c_out``, ``vn = v . n``, ``n`` the unit vector inner -> outer, and a zero
    ghost state on inflow walls.  Written in the device operator's difference form
    ``(A x)_i = sum_f w_if (x_other - x_i) + diag_extra_i x_i``; apply with ``alpha = 1, beta = 0``.
    Returns ``(w_inner[F], w_outer[F], diag_extra[n_cells])``.
    """
    n = g.n_cells
    vel = np.asarray(vel, np.float64)[: g.dim]
    coef, b_coef = face_coefficients(g)
    d = g.center[g.outer] - g.center[g.inner]
    dist = g.area / coef
    vn = (d @ vel) / dist
    a_in = g.area / g.volume[g.inner]
    a_out = g.area / g.volume[g.outer]
    w_inner = -nu * coef / g.volume[g.inner] + a_in * np.minimum(vn, 0.0)
    w_outer = -nu * coef / g.volume[g.outer] - a_out * np.maximum(vn, 0.0)
    diag = np.zeros(g.n_total)
    np.add.at(diag, g.inner, a_in * vn)
    np.add.at(diag, g.outer, -a_out * vn)
    if g.n_bfaces:
        db = g.b_center - g.center[g.b_cell]
        bdist = g.b_area / b_coef
        bvn = (db @ vel) / bdist
        np.add.at(diag, g.b_cell, nu * b_coef / g.volume[g.b_cell]
                  + (g.b_area / g.volume[g.b_cell]) * np.maximum(bvn, 0.0))
    return w_inner, w_outer, diag[:n].copy()


def _box_faces(nx: int, ny: int, nz: int, k0: int, k1: int, nz_glob: int,
               hx: float, hy: float, hz: float):
    """Faces of the cells with global k in [k0, k1) of an nx*ny*nz_glob box.

    Returns global-id face arrays in cell-major (+x,+y,+z) order for every face whose inner
    (lower-id) cell lies in the slab, plus the -z interface faces whose *outer* cell lies in
    the slab (inner cell in the slab below), placed first so a global face order restricted
    to "faces touching the slab" is preserved: those faces have lower inner ids.
    """
    nloc = k1 - k0
    n = nx * ny * nloc
    ids = np.arange(n, dtype=np.int64) + np.int64(k0) * nx * ny
    i = ids % nx
    j = (ids // nx) % ny
    k = ids // (nx * ny)
    has_x = i < nx - 1
    has_y = j < ny - 1
    has_z = k < nz_glob - 1
    cnt = has_x.astype(np.int64) + has_y + has_z
    off = np.cumsum(cnt) - cnt
    nf = int(cnt.sum())
    inner = np.empty(nf, np.int64)
    outer = np.empty(nf, np.int64)
    axis = np.empty(nf, np.int8)
    px = off[has_x]
    inner[px] = ids[has_x]
    outer[px] = ids[has_x] + 1
    axis[px] = 0
    py = (off + has_x)[has_y]
    inner[py] = ids[has_y]
    outer[py] = ids[has_y] + nx
    axis[py] = 1
    pz = (off + has_x + has_y)[has_z]
    inner[pz] = ids[has_z]
    outer[pz] = ids[has_z] + nx * ny
    axis[pz] = 2
    if k0 > 0:  # interface faces owned by the slab below (their +z faces)
        low = np.arange(nx * ny, dtype=np.int64) + np.int64(k0 - 1) * nx * ny
        inner = np.concatenate([low, inner])
        outer = np.concatenate([low + nx * ny, outer])
        axis = np.concatenate([np.full(nx * ny, 2, np.int8), axis])
    areas = np.array([hy * hz, hx * hz, hx * hy])
    area = areas[axis]

    # Boundary (wall) faces, cell-major, order -x,+x,-y,+y,-z,+z.
    walls = [i == 0, i == nx - 1, j == 0, j == ny - 1, k == 0, k == nz_glob - 1]
    wcnt = np.zeros(n, np.int64)
    for w in walls:
        wcnt += w
    woff = np.cumsum(wcnt) - wcnt
    nb = int(wcnt.sum())
    b_cell = np.empty(nb, np.int64)
    b_side = np.empty(nb, np.int8)
    run = woff.copy()
    for s, w in enumerate(walls):
        p = run[w]
        b_cell[p] = ids[w]
        b_side[p] = s
        run = run + w
    return inner, outer, area, b_cell, b_side


def _centers(ids: np.ndarray, nx: int, ny: int, hx: float, hy: float, hz: float) -> np.ndarray:
    i = ids % nx
    j = (ids // nx) % ny
    k = ids // (nx * ny)
    return np.stack([(i + 0.5) * hx, (j + 0.5) * hy, (k + 0.5) * hz], axis=1)


def _wall_centers(cc: np.ndarray, b_side: np.ndarray, hx: float, hy: float, hz: float) -> np.ndarray:
    bc = cc.copy()
    half = np.array([hx, hy, hz]) * 0.5
    for s in range(6):
        m = b_side == s
        ax = s // 2
        bc[m, ax] += half[ax] if (s % 2) else -half[ax]
    return bc


def structured_box(nx: int, ny: Optional[int] = None, nz: Optional[int] = None,
                   lengths: Sequence[float] = (1.0, 1.0, 1.0),
                   dirichlet: bool = True) -> FaceGraph:
    """The N^3 structured-as-unstructured Poisson mesh of SURVEY.md 8d.

    Unit cube by default, h = 1/n; cell volume h^3; interior faces of area h^2 at centre
    distance h; with ``dirichlet`` the 6 walls carry boundary faces of area h^2 whose face
    centre is h/2 from the cell centre (ghost value 0 there), which makes ``-L`` SPD with
    off-diagonals -1/h^2 and diagonal (6 + #walls)/h^2.
    """
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    return structured_box_slab(nx, ny, nz, 0, nz, lengths, dirichlet)


def structured_box_slab(nx: int, ny: int, nz_glob: int, k0: int, k1: int,
                        lengths: Sequence[float] = (1.0, 1.0, 1.0),
                        dirichlet: bool = True, rank_of_k=None) -> FaceGraph:
    """Local face graph of the z-slab ``k0 <= k < k1`` of an nx*ny*nz_glob box.

    Owned cells are numbered 0..n_owned-1 in global-id order; halo cells (the plane below,
    then the plane above) follow, each plane in global-id order.  ``global_id`` maps local ->
    global ids.  This is the direct generator for the row partition of SURVEY.md 8e; it
    agrees with :func:`stormruler_amd.partition.partition_graph` applied to the global mesh
    (tests/test_partition.py) without ever materialising the global mesh.
    """
    hx, hy, hz = lengths[0] / nx, lengths[1] / ny, lengths[2] / nz_glob
    inner_g, outer_g, area, b_cell_g, b_side = _box_faces(nx, ny, nz_glob, k0, k1, nz_glob, hx, hy, hz)
    plane = nx * ny
    n_owned = plane * (k1 - k0)
    base = np.int64(k0) * plane
    halo_ids = []
    halo_owner = []
    if k0 > 0:
        halo_ids.append(np.arange(plane, dtype=np.int64) + np.int64(k0 - 1) * plane)
        halo_owner.append(np.full(plane, -1 if rank_of_k is None else rank_of_k(k0 - 1), np.int32))
    if k1 < nz_glob:
        halo_ids.append(np.arange(plane, dtype=np.int64) + np.int64(k1) * plane)
        halo_owner.append(np.full(plane, -1 if rank_of_k is None else rank_of_k(k1), np.int32))
    halo = np.concatenate(halo_ids) if halo_ids else np.zeros(0, np.int64)
    owner = np.concatenate(halo_owner) if halo_owner else np.zeros(0, np.int32)
    gid = np.concatenate([np.arange(n_owned, dtype=np.int64) + base, halo])

    def to_local(g: np.ndarray) -> np.ndarray:
        loc = g - base
        below = g < base
        above = g >= base + n_owned
        if k0 > 0:
            loc = np.where(below, n_owned + (g - (base - plane)), loc)
        off_above = n_owned + (plane if k0 > 0 else 0)
        loc = np.where(above, off_above + (g - (base + n_owned)), loc)
        return loc.astype(np.int64)

    cc = _centers(gid, nx, ny, hx, hy, hz)
    vol = np.full(gid.shape[0], hx * hy * hz)
    if dirichlet:
        b_cell = to_local(b_cell_g)
        b_area = np.array([hy * hz, hy * hz, hx * hz, hx * hz, hx * hy, hx * hy])[b_side]
        b_center = _wall_centers(cc[b_cell], b_side, hx, hy, hz)
    else:
        b_cell = np.zeros(0, np.int64)
        b_area = np.zeros(0)
        b_center = np.zeros((0, 3))
    g = FaceGraph(n_cells=n_owned, dim=3, inner=to_local(inner_g), outer=to_local(outer_g),
                  area=area, center=cc, volume=vol, b_cell=b_cell, b_area=b_area,
                  b_center=b_center, n_halo=int(halo.shape[0]), global_id=gid, halo_owner=owner)
    return g


# ---------------------------------------------------------------------------------------
# Cell orderings.  The reference exposes an entity permutation hook
# (source/Storm/Mallard/MeshUnstructured.hpp:443-459, 557-612); METIS is not available in
# this image (nor used by the reference, CMakeLists.txt:373-384), so the orderings below
# are the build's own.  ``perm[new] = old``.


def random_permutation(n: int, seed: int = 12345) -> np.ndarray:
    """The seeded scramble of SURVEY.md 8d ("unstructured stress variant")."""
    return np.random.default_rng(seed).permutation(n).astype(np.int64)


def jitter_geometry(g: FaceGraph, h: float, frac: float = 0.2, seed: int = 2024) -> FaceGraph:
    """The same face graph with a perturbed geometry: every cell centre displaced by at most ``frac * h`` per coordinate
    and every face area scaled by 1 +- frac / 2 (seeded).  Volumes stay: the operator remains symmetric positive
    definite, but no two of its weights ``A_f / (d_f V_i)`` are equal any more -- what a Triangle / TetGen mesh of the
    reference looks like to the record formats (fp64 weights), at any size."""
    rng = np.random.default_rng(seed)
    center = g.center.copy()
    center[: g.n_cells] += (frac * h) * (2.0 * rng.random((g.n_cells, g.dim)) - 1.0)
    area = g.area * (1.0 + (0.5 * frac) * (2.0 * rng.random(g.n_faces) - 1.0))
    return FaceGraph(n_cells=g.n_cells, dim=g.dim, inner=g.inner, outer=g.outer, area=area, center=center,
                     volume=g.volume, b_cell=g.b_cell, b_area=g.b_area, b_center=g.b_center, n_halo=g.n_halo,
                     global_id=g.global_id, halo_owner=g.halo_owner)


def tile_ordering(nx: int, ny: int, nz: int, ty: int = 16, tz: int = 16) -> np.ndarray:
    """Pencil-tile ordering: x-lines grouped into (ty x tz) tiles of the y-z plane.

    Keeps every x-line contiguous (so the +-1 and the in-tile +-nx, +-nx*ny gathers of a
    wavefront stay unit-stride) while shrinking the reuse distance of the z-neighbours from
    nx*ny rows to ty*nx rows, so they are served by the XCD's 4 MiB L2.
    """
    j, k = np.meshgrid(np.arange(ny), np.arange(nz), indexing="xy")  # shape [nz, ny]
    key = ((k // tz) * ((ny + ty - 1) // ty) + (j // ty)) * (ty * tz) + (k % tz) * ty + (j % ty)
    order = np.argsort(key.ravel(), kind="stable")  # line index (k*ny + j) in tile order
    lines = order.astype(np.int64)
    return (lines[:, None] * nx + np.arange(nx, dtype=np.int64)[None, :]).ravel()


def rcm_ordering(g: FaceGraph) -> np.ndarray:
    """Reverse Cuthill-McKee on the owned-cell graph (bandwidth reduction for gathers).

    Small graphs use the build's own BFS below; past 200k cells the same algorithm from
    ``scipy.sparse.csgraph`` (compiled) is used when scipy is importable -- host preprocessing in the
    role METIS plays in the north star (METIS itself is not in this image)."""
    n = g.n_cells
    m = (g.inner < n) & (g.outer < n)
    if n > 200_000:
        try:
            import scipy.sparse as sp
            from scipy.sparse.csgraph import reverse_cuthill_mckee

            i, o = g.inner[m], g.outer[m]
            adj = sp.coo_matrix((np.ones(2 * i.size, np.int8), (np.concatenate([i, o]), np.concatenate([o, i]))),
                                shape=(n, n)).tocsr()
            return reverse_cuthill_mckee(adj, symmetric_mode=True).astype(np.int64)
        except ImportError:  # pragma: no cover
            pass
    # BFS from a minimum-degree vertex of each component, neighbours by increasing degree.
    a = np.concatenate([g.inner[m], g.outer[m]])
    b = np.concatenate([g.outer[m], g.inner[m]])
    order = np.argsort(a, kind="stable")
    a, b = a[order], b[order]
    ptr = np.zeros(n + 1, np.int64)
    np.add.at(ptr, a + 1, 1)
    ptr = np.cumsum(ptr)
    deg = np.diff(ptr)
    visited = np.zeros(n, bool)
    out = np.empty(n, np.int64)
    pos = 0
    by_deg = np.argsort(deg, kind="stable")
    for start in by_deg:
        if visited[start]:
            continue
        visited[start] = True
        out[pos] = start
        head, pos = pos, pos + 1
        while head < pos:
            v = out[head]
            head += 1
            nb = b[ptr[v]:ptr[v + 1]]
            nb = nb[~visited[nb]]
            if nb.size:
                nb = np.unique(nb)
                nb = nb[np.argsort(deg[nb], kind="stable")]
                visited[nb] = True
                out[pos:pos + nb.size] = nb
                pos += nb.size
    return out[::-1].copy()


def geometric_ordering(g: FaceGraph, mode: str = "auto") -> Tuple[np.ndarray, str]:
    """The library's ordering from the cell centres (``storm_hip_order_cells``, native and threaded: 0.3 - 0.5 s at
    256^3 where scipy's reverse Cuthill-McKee takes 6 - 11 s): ``"auto"`` returns the lexicographic order of a lattice
    where the centres form a tensor-product grid -- a renumbered structured mesh gets its natural order, and with it
    the lattice record formats, back -- and the Z-order (Morton) curve of the centres otherwise; ``"morton"`` forces
    the curve (what a Triangle / TetGen mesh gets).  Returns ``(order, kind)``: new cell ``i`` is old cell ``order[i]``
    (the argument of ``permute_cells``), ``kind`` is ``"lattice"`` or ``"morton"``.  No device is touched."""
    import ctypes as C

    from ._lib import check, lib

    n = g.n_cells
    centers = np.ascontiguousarray(g.center[:n], dtype=np.float64)
    order = np.empty(n, np.int64)
    kind = C.c_int32(0)
    check(lib.storm_hip_order_cells(g.dim, n, centers.ctypes.data_as(C.POINTER(C.c_double)),
                                    {"auto": 0, "morton": 1, "lattice": 2, "hilbert": 3}[mode],
                                    order.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(kind)))
    return order, {1: "lattice", 2: "morton", 3: "hilbert"}.get(kind.value, "none")


def permute_cells(g: FaceGraph, perm: np.ndarray) -> FaceGraph:
    """Renumber owned cells: new cell ``i`` is old cell ``perm[i]`` (halo cells keep their slot).

    Faces keep their order and their inner/outer roles, exactly like the reference's
    ``permute`` hook renumbers entities without touching the adjacency rows' order.
    """
    n = g.n_cells
    assert perm.shape == (n,)
    inv = np.empty(g.n_total, np.int64)
    inv[perm] = np.arange(n, dtype=np.int64)
    inv[n:] = np.arange(n, g.n_total, dtype=np.int64)
    full = np.concatenate([perm, np.arange(n, g.n_total, dtype=np.int64)])
    return FaceGraph(n_cells=n, dim=g.dim, inner=inv[g.inner], outer=inv[g.outer], area=g.area,
                     center=g.center[full], volume=g.volume[full], b_cell=inv[g.b_cell],
                     b_area=g.b_area, b_center=g.b_center, n_halo=g.n_halo,
                     global_id=None if g.global_id is None else g.global_id[full],
                     halo_owner=g.halo_owner)


def assemble_csr(g: FaceGraph, alpha: float, beta: float):
    """Assembled matrix of ``y = beta*x + alpha*L(x)`` on owned rows (host check helper).

    Returns scipy CSR of shape [n_cells, n_total].  Used by tests as an independent
    cross-check of the oracle's matrix-free face loop; not used by the device path.
    """
    import scipy.sparse as sp

    n, nt = g.n_cells, g.n_total
    coef, b_coef = face_coefficients(g)
    wi = coef / g.volume[g.inner]
    wo = coef / g.volume[g.outer]
    rows = np.concatenate([g.inner, g.inner, g.outer, g.outer])
    cols = np.concatenate([g.outer, g.inner, g.inner, g.outer])
    vals = np.concatenate([wi, -wi, wo, -wo]) * alpha
    if g.n_bfaces:
        rows = np.concatenate([rows, g.b_cell])
        cols = np.concatenate([cols, g.b_cell])
        vals = np.concatenate([vals, -alpha * b_coef / g.volume[g.b_cell]])
    rows = np.concatenate([rows, np.arange(n)])
    cols = np.concatenate([cols, np.arange(n)])
    vals = np.concatenate([vals, np.full(n, beta)])
    keep = rows < n
    a = sp.coo_matrix((vals[keep], (rows[keep], cols[keep])), shape=(n, nt)).tocsr()
    a.sum_duplicates()
    return a


def periodic_z_local_graph(nx, ny, nz):
    """One rank's local graph of a box that is PERIODIC in z, the rank its own neighbour: owned = the whole box without its
    z walls; halo = [copy of plane nz - 1 (below plane 0), copy of plane 0 (above plane nz - 1)], positioned at their periodic
    images.  Returns (graph, send_idx): with ``set_halo([0], [0, n_halo], send_idx, [0, n_halo])`` on a communicator of size 1
    the rank exchanges both planes with itself -- the multi-rank code path with the network taken out
    (tests/test_gpu_comm.py, tools/comm_path_overhead.py, bench.py's ``multi_rank_path_at_one_rank``)."""
    g = structured_box(nx, ny, nz)
    P, N = nx * ny, nx * ny * nz
    keep = ~np.isclose(np.abs(g.b_center[:, 2] - 0.5), 0.5)  # drop the z = 0 and z = 1 wall faces
    bottom, top = np.arange(P, dtype=np.int64), np.arange(N - P, N, dtype=np.int64)
    halo_a = N + np.arange(P, dtype=np.int64)       # images of the top plane, below the bottom plane
    halo_b = N + P + np.arange(P, dtype=np.int64)   # images of the bottom plane, above the top plane
    inner = np.concatenate([g.inner, halo_a, top])
    outer = np.concatenate([g.outer, bottom, halo_b])
    area = np.concatenate([g.area, np.full(2 * P, g.area[-1])])
    ca, cb = g.center[top].copy(), g.center[bottom].copy()
    ca[:, 2] -= 1.0
    cb[:, 2] += 1.0
    loc = FaceGraph(n_cells=N, dim=3, inner=inner, outer=outer, area=area,
                    center=np.concatenate([g.center, ca, cb]), volume=np.concatenate([g.volume, g.volume[:2 * P]]),
                    b_cell=g.b_cell[keep], b_area=g.b_area[keep], b_center=g.b_center[keep], n_halo=2 * P,
                    global_id=np.concatenate([np.arange(N), top, bottom]),
                    halo_owner=np.zeros(2 * P, np.int32))
    loc.validate()
    return loc, np.concatenate([top, bottom])
