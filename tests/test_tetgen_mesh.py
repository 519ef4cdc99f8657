"""The 3-D branch of the Triangle / TetGen reader (Mallard/IoTetgen.hpp:60-100, 139-175, 207-209)
and a genuinely unstructured 3-D workload: a seeded tetrahedral box (six shapes of cells, all weights distinct, rows
of 2 - 4 neighbours).  The numpy restatement (stormruler_amd/io_tetgen.py) is checked against closed forms and the
oracle's face loop against the assembled matrix; the library's native reader (csrc/mesh_host.hip) against the numpy
restatement array for array; the HIP path against the oracle (-m gpu)."""
import os

import numpy as np
import pytest

from oracle import oracle
from stormruler_amd import host_mesh, io_tetgen, mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("inner", "outer", "area", "center", "volume", "b_cell", "b_area", "b_center")


def _same(g, h):
    for k in KEYS:
        a, b = getattr(g, k), getattr(h, k)
        assert a.shape == b.shape and np.array_equal(a, b), k


def _box(n, **kw):
    pos, bf, cells = io_tetgen.tet_box(n, **kw)
    return pos, bf, np.ones(len(bf), np.int64), cells


@pytest.mark.parametrize("n", [1, 2, 5])
def test_tet_box_counts_and_geometry(n):
    pos, bf, lab, cells = _box(n)
    g = io_tetgen.face_graph_from_simplices(pos, bf, lab, cells)
    assert (g.n_cells, g.n_bfaces, g.dim) == (6 * n ** 3, 12 * n * n, 3)
    assert 4 * g.n_cells == 2 * g.n_faces + g.n_bfaces  # every cell has four sides
    assert abs(g.volume.sum() - 1.0) < 1e-13 and abs(g.b_area.sum() - 6.0) < 1e-13
    assert np.all(g.inner < g.outer)  # inner = the first inserted owner
    p = [pos[cells[:, k]] for k in range(4)]
    det = np.einsum("ij,ij->i", p[1] - p[0], np.cross(p[2] - p[0], p[3] - p[0]))
    assert det.min() > 0 and np.allclose(det / 6.0, g.volume, rtol=1e-13)  # positively oriented, no cell inverted
    # Euler characteristic of a ball: V - E + F - C = 1
    e = np.sort(np.concatenate([cells[:, [a, b]] for a, b in ((0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3))]), axis=1)
    n_edges = np.unique(e[:, 0] * pos.shape[0] + e[:, 1]).size
    assert pos.shape[0] - n_edges + (g.n_faces + g.n_bfaces) - g.n_cells == 1
    # a tetrahedron's four side areas, as vectors with outward normals, sum to zero: check through the face graph
    # that every interior face's area is the same seen from both of its cells
    for f in (0, g.n_faces // 2, g.n_faces - 1):
        shared = np.intersect1d(cells[g.inner[f]], cells[g.outer[f]])
        assert shared.size == 3
        q = pos[shared]
        assert abs(0.5 * np.linalg.norm(np.cross(q[1] - q[0], q[2] - q[0])) - g.area[f]) <= 1e-15


def test_unit_tetrahedron_known_answers(tmp_path):
    """One cell, everything by hand: volume 1/6, centre (1/4, 1/4, 1/4), side areas 1/2, 1/2, 1/2, sqrt(3)/2 in the
    order of Tetrahedron::faces() (Shape.hpp:590-594)."""
    (tmp_path / "t.1.node").write_text("# unit tetrahedron\n4 3 0 0\n0 0 0 0\n1 1 0 0\n2 0 1 0\n3 0 0 1\n")
    (tmp_path / "t.1.edge").write_text("0 1\n")
    (tmp_path / "t.1.face").write_text("0 1\n")  # TetGen "may not generate all the faces": none listed ...
    (tmp_path / "t.1.ele").write_text("1 4 0\n0 0 1 2 3\n")
    with pytest.raises(RuntimeError, match="unlabelled side has a single adjacent cell"):
        io_tetgen.read_tetgen(str(tmp_path / "t.1"), 3)  # ... so its four sides get label 0 with one cell each
    (tmp_path / "t.1.face").write_text("4 1\n0 0 2 1 7\n1 0 1 3 7\n2 1 2 3 7  # the slanted one\n3 2 0 3 7\n")
    for g in (io_tetgen.read_tetgen(str(tmp_path / "t.1"), 3),
              host_mesh.HostMesh.read_tetgen(str(tmp_path / "t.1."), 3).face_graph()):
        assert (g.n_cells, g.n_faces, g.n_bfaces) == (1, 0, 4)
        assert g.volume[0] == 1.0 / 6.0 and np.array_equal(g.center[0], [0.25, 0.25, 0.25])
        assert np.allclose(g.b_area, [0.5, 0.5, np.sqrt(3.0) / 2.0, 0.5], rtol=1e-15)
        assert np.allclose(g.b_center[2], [1 / 3, 1 / 3, 1 / 3], rtol=1e-15)


def test_two_tetrahedra_inner_outer_and_orientation(tmp_path):
    base = "5 3 0 0\n0 0 0 0\n1 1 0 0\n2 0 1 0\n3 0 0 1\n4 1 1 1\n"
    (tmp_path / "t.1.node").write_text(base)
    (tmp_path / "t.1.edge").write_text("0 0\n")
    bnd = [(0, 2, 1), (0, 1, 3), (2, 0, 3), (1, 2, 4), (2, 3, 4), (3, 1, 4)]
    (tmp_path / "t.1.face").write_text(f"{len(bnd)} 1\n" + "".join(f"{i} {a} {b} {c} 1\n" for i, (a, b, c) in enumerate(bnd)))
    (tmp_path / "t.1.ele").write_text("2 4 0\n0 0 1 2 3\n1 2 1 3 4 # two nodes swapped\n")
    with pytest.raises(RuntimeError, match="second cell cannot be the outer one"):
        io_tetgen.read_tetgen(str(tmp_path / "t.1"))  # cell 1 has the other orientation: both cells see (1, 2, 3) alike
    with pytest.raises(RuntimeError, match="second cell cannot be the outer one"):
        host_mesh.HostMesh.read_tetgen(str(tmp_path / "t.1"))
    (tmp_path / "t.1.ele").write_text("2 4 0\n0 0 1 2 3\n1 1 2 3 4\n")
    g = io_tetgen.read_tetgen(str(tmp_path / "t.1"))
    _same(g, host_mesh.HostMesh.read_tetgen(str(tmp_path / "t.1")).face_graph())
    assert (g.n_faces, g.n_bfaces) == (1, 6) and (g.inner[0], g.outer[0]) == (0, 1)
    assert abs(g.area[0] - np.sqrt(3.0) / 2.0) < 1e-15


def test_listed_interior_faces_come_first(tmp_path):
    """assign_labels (MeshUnstructured.hpp:464-500) stable-sorts by label: a listed face with marker 0 precedes the faces
    created by the cells, whatever cell owns it."""
    pos, bf, lab, cells = _box(2)
    g0 = io_tetgen.face_graph_from_simplices(pos, bf, lab, cells)
    last = np.array([np.intersect1d(cells[g0.inner[-1]], cells[g0.outer[-1]])])  # the last created face, listed first
    g1 = io_tetgen.face_graph_from_simplices(pos, np.concatenate([last, bf]), np.concatenate([[0], lab]), cells)
    assert (g1.inner[0], g1.outer[0]) == (g0.inner[-1], g0.outer[-1])
    assert np.array_equal(g1.inner[1:], g0.inner[:-1]) and np.array_equal(g1.b_cell, g0.b_cell)
    m = host_mesh.HostMesh.from_simplices(pos, np.concatenate([last, bf]), np.concatenate([[0], lab]), cells)
    _same(g1, m.face_graph())


@pytest.mark.parametrize("n", [3, 9])
def test_native_reader_equals_the_numpy_restatement(n, tmp_path):
    pos, bf, lab, cells = _box(n)
    g = io_tetgen.face_graph_from_simplices(pos, bf, lab, cells)
    _same(g, host_mesh.HostMesh.from_simplices(pos, bf, lab, cells).face_graph())
    # files: the library's writer / reader and the numpy writer / reader, crosswise
    host_mesh.write_tetgen(str(tmp_path / "a.1"), pos, bf, lab, cells)
    io_tetgen.write_tetgen(str(tmp_path / "b.1"), pos, bf, lab, cells, comment="numpy writer")
    for name in ("a.1", "b.1"):
        _same(g, host_mesh.HostMesh.read_tetgen(str(tmp_path / name) + ".", 3).face_graph())
        _same(g, io_tetgen.read_tetgen(str(tmp_path / name), 3))


def test_native_reader_on_the_reference_2d_mesh(golden):
    u = golden["baseline_md_probe"]["unstructured"]
    g = io_tetgen.read_triangle(os.path.join(ROOT, u["mesh"]))
    _same(g, host_mesh.HostMesh.read_tetgen(os.path.join(ROOT, u["mesh"]), 2).face_graph())


@pytest.mark.parametrize("name,nodes,cells,edges", [("rectangle.1", 6725, 12776, 19500), ("step.1", 40303, 79672, 119974)])
def test_native_reader_on_the_references_other_2d_meshes(name, nodes, cells, edges, tmp_path):
    """rectangle.1.* and step.1.* (tests/_data/mesh of the reference; stored gzip-compressed here): the native reader
    against the numpy restatement array for array, the counts of the files' headers, Euler's formula for a triangulated
    region with holes (V - E + F = 1 - holes) and the area as the sum of the cells'."""
    import gzip
    import shutil

    for e in ("node", "ele", "edge"):
        with gzip.open(os.path.join(ROOT, "tests", "golden", "mesh", f"{name}.{e}.gz"), "rb") as src, open(tmp_path / f"{name}.{e}", "wb") as dst:
            shutil.copyfileobj(src, dst)
    g = io_tetgen.read_triangle(str(tmp_path / name) + ".")
    m = host_mesh.HostMesh.read_tetgen(str(tmp_path / name) + ".", 2)
    _same(g, m.face_graph())
    assert g.n_cells == cells and g.n_faces + g.n_bfaces == edges
    holes = 1 - (nodes - edges + cells)
    assert holes in (0, 1, 2)  # (rectangle: a plain rectangle; step: the channel with its step is still simply connected)
    assert np.all(g.volume > 0) and np.all(g.area > 0) and np.all(g.inner != g.outer)
    x = np.sin(0.37 * np.arange(g.n_cells))
    gn = mesh.FaceGraph(g.n_cells, 2, g.inner, g.outer, g.area, g.center, g.volume)
    y = oracle.StencilOperator(gn, 1.0, 0.0).apply(x)
    assert abs(np.dot(g.volume, y)) <= 1e-11 * np.abs(g.volume * y).sum()  # the Neumann operator conserves


def test_reader_errors(tmp_path):
    pos, bf, lab, cells = _box(2)
    io_tetgen.write_tetgen(str(tmp_path / "b.1"), pos, bf, lab, cells)
    for read in (lambda p, d: io_tetgen.read_tetgen(p, d), lambda p, d: host_mesh.HostMesh.read_tetgen(p, d)):
        with pytest.raises(RuntimeError, match="Unexpected number of the dimensions"):
            read(str(tmp_path / "b.1"), 2)  # mesh_dim_v<Mesh> = 2 against a 3-D node file (IoTetgen.hpp:70-74)
        with pytest.raises(RuntimeError, match="Cannot open"):
            read(str(tmp_path / "nope.1"), 3)
    os.remove(tmp_path / "b.1.face")
    with pytest.raises(RuntimeError, match="Cannot open"):
        io_tetgen.read_tetgen(str(tmp_path / "b.1"), 3)  # the 3-D branch needs the .face file (IoTetgen.hpp:141-145)
    with pytest.raises(RuntimeError, match="Cannot open the face file"):
        host_mesh.HostMesh.read_tetgen(str(tmp_path / "b.1"), 3)
    (tmp_path / "b.1.face").write_text("3 1\n0 0 1 2 1\n")  # short
    with pytest.raises(RuntimeError, match="Cannot read the faces"):
        io_tetgen.read_tetgen(str(tmp_path / "b.1"), 3)
    with pytest.raises(RuntimeError, match="Cannot read the faces"):
        host_mesh.HostMesh.read_tetgen(str(tmp_path / "b.1"), 3)
    (tmp_path / "b.1.ele").write_text("1 3 0\n0 0 1 2\n")
    (tmp_path / "b.1.face").write_text("0 1\n")
    with pytest.raises(RuntimeError, match="Unexpected number of the nodes per cell"):
        host_mesh.HostMesh.read_tetgen(str(tmp_path / "b.1"), 3)


def test_native_reader_rejects_headers_that_lie(tmp_path):
    """Counts no file of that size can hold (2^62: the size check must not overflow), fractions, NaN, 1e300 in a header:
    an error, not a crash (found by fuzzing the reader under ASan / UBSan: 2 400 mutated file sets, tools-free, on the CPU)."""
    pos, bf, lab, cells = _box(2)
    prefix = str(tmp_path / "b.1")
    io_tetgen.write_tetgen(prefix, pos, bf, lab, cells)
    good = {e: open(prefix + "." + e).read() for e in ("node", "ele", "face", "edge")}

    def with_header(ext, header):
        for e, text in good.items():
            open(prefix + "." + e, "w").write(text)
        lines = good[ext].split("\n")
        lines[0] = header
        open(prefix + "." + ext, "w").write("\n".join(lines))

    cases = [("node", "4611686018427387904 3 0 0", "header"), ("node", "4503599627370496 3 0 0", "Cannot read the nodes"), ("node", "nan 3 0 0", "header"),
             ("node", "27.5 3 0 0", "header"), ("node", "27 3 1e300 0", "header"), ("node", "-27 3 0 0", "header"),
             ("node", "27 3 4503599627370496 0", "Cannot read the nodes"),
             ("ele", "4611686018427387904 4 0", "Cannot read the cells"), ("ele", "-1 4 0", "Cannot read the cells"),
             ("face", "4611686018427387904 1", "Cannot read the faces"), ("edge", "3074457345618258603 1", "Cannot read the edges")]
    for ext, header, message in cases:
        with_header(ext, header)
        with pytest.raises(RuntimeError, match=message):
            host_mesh.HostMesh.read_tetgen(prefix, 3)
    with_header("node", good["node"].split("\n")[0])
    assert host_mesh.HostMesh.read_tetgen(prefix, 3).view().n_cells == len(cells)


def test_oracle_face_loop_on_tetrahedra_against_the_assembled_matrix():
    pos, bf, lab, cells = _box(6)
    g = io_tetgen.face_graph_from_simplices(pos, bf, lab, cells)
    x = np.sin(0.37 * np.arange(g.n_cells))
    for alpha, beta in ((-1.0, 0.0), (-1e-2, 1.0)):
        y = oracle.StencilOperator(g, alpha, beta).apply(x)
        ya = mesh.assemble_csr(g, alpha, beta) @ x
        assert np.abs(y - ya).max() <= 1e-13 * np.abs(ya).max()
    # the Neumann operator (the reference's: boundary faces skipped, Playground.cpp:119) conserves: sum V_i (L x)_i = 0
    gn = mesh.FaceGraph(g.n_cells, 3, g.inner, g.outer, g.area, g.center, g.volume)
    yn = oracle.StencilOperator(gn, 1.0, 0.0).apply(x)
    assert abs(np.dot(g.volume, yn)) <= 1e-12 * np.abs(g.volume * yn).sum()


def test_permute_and_morton_order_keep_the_operator():
    pos, bf, lab, cells = _box(5)
    m = host_mesh.HostMesh.from_simplices(pos, bf, lab, cells)
    g = m.face_graph()
    assert m.order_cells("morton") == "morton"
    gp = m.face_graph()
    order = gp.global_id
    _same(mesh.permute_cells(g, order), gp)
    x = np.cos(0.11 * np.arange(g.n_cells))
    y = oracle.StencilOperator(g, -1.0, 0.0).apply(x)
    yp = oracle.StencilOperator(gp, -1.0, 0.0).apply(x[order])
    assert np.array_equal(yp, y[order])  # faces keep their order: the very same sums


@pytest.mark.parametrize("n_parts", [2, 5])
def test_partition_of_a_renumbered_mesh_speaks_the_original_ids(n_parts):
    """A mesh renumbered along the Hilbert curve and then partitioned natively: the parts' global ids are the
    file's, what rank r sends to q is in order what q's halo group expects, and the partitioned apply is the
    global one.  (Halo groups sorted by position in the renumbered mesh would not pair with send lists sorted by id.)"""
    pos, bf, lab, cells = _box(5)
    m = host_mesh.HostMesh.from_simplices(pos, bf, lab, cells)
    g = m.face_graph()
    assert m.order_cells("hilbert") == "hilbert"
    center = m.face_graph().center
    part = host_mesh.partition_rcb(center, n_parts)
    locs = [m.partition(part, n_parts, r) for r in range(n_parts)]
    graphs = [l.face_graph() for l in locs]
    plans = [l.halo_plan() for l in locs]
    x = np.sin(0.37 * np.arange(g.n_cells))
    y_glob = oracle.StencilOperator(g, -1e-2, 1.0).apply(x)
    y = np.full_like(x, np.nan)
    for r in range(n_parts):
        pr, lr = plans[r], graphs[r]
        for qi, q in enumerate(pr.nbr_rank):
            sent = lr.global_id[pr.send_idx[pr.send_ptr[qi]:pr.send_ptr[qi + 1]]]
            pq, lq = plans[q], graphs[q]
            j = list(pq.nbr_rank).index(r)
            assert np.array_equal(sent, lq.global_id[lq.n_cells + pq.recv_ptr[j]: lq.n_cells + pq.recv_ptr[j + 1]])
        yl = oracle.StencilOperator(lr, -1e-2, 1.0).apply(x[lr.global_id])
        y[lr.global_id[: lr.n_cells]] = yl[: lr.n_cells]
    assert np.abs(y - y_glob).max() <= 1e-13 * np.abs(y_glob).max()


# ---- the HIP path ------------------------------------------------------------------------------------------------


@pytest.mark.gpu
@pytest.mark.parametrize("ordering", ["file", "morton", "random"])
def test_hip_spmv_and_cg_on_a_tetrahedral_mesh(ordering):
    """<= 20 k cells: SpMV against the oracle's face loop to 1e-13, CG against the oracle's solve."""
    from stormruler_amd import api

    pos, bf, lab, cells = _box(14)  # 16 464 cells
    m = host_mesh.HostMesh.from_simplices(pos, bf, lab, cells)
    if ordering == "morton":
        assert m.order_cells("morton") == "morton"
    elif ordering == "random":
        m.permute_cells(mesh.random_permutation(6 * 14 ** 3))
    g = m.face_graph()
    ctx = api.Context(0)
    mat = m.create_operator(ctx)  # storm_hip_op_create_from_mesh_object
    st = mat.stats()
    assert st["max_row_len"] == 4 and st["value_dictionary_size"] == 0 and st["tail_nnz"] == 0
    assert st["nnz_offdiag"] == 2 * g.n_faces
    x_host = np.sin(0.37 * np.arange(g.n_cells))
    x, y = api.DeviceVector.from_numpy(ctx, x_host), api.DeviceVector(ctx, g.n_cells)
    for alpha, beta in ((-1.0, 0.0), (-1e-3, 1.0)):
        mat.apply(alpha, beta, x, y)
        y_ref = oracle.StencilOperator(g, alpha, beta).apply(x_host)
        assert np.abs(y.to_numpy() - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    b, xs = api.DeviceVector(ctx, g.n_cells), api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    s = api.CgSolver()
    assert s.solve(xs, b, api.HipStencilOperator(mat, -1.0, 0.0))
    ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells))
    assert ref.converged and abs(s.iteration - ref.iterations) <= max(2, int(0.02 * ref.iterations))
    assert np.linalg.norm(xs.to_numpy() - ref.x) <= 1e-8 * np.linalg.norm(ref.x)
    # ... and the same mesh through the face-graph constructor gives the same operator, bit for bit
    mat2 = api.StencilMatrix.from_face_graph(ctx, g)
    y2 = api.DeviceVector(ctx, g.n_cells)
    mat2.apply(-1.0, 0.0, x, y2)
    mat.apply(-1.0, 0.0, x, y)
    assert np.array_equal(y.to_numpy(), y2.to_numpy())
    ctx.close()


@pytest.mark.gpu
def test_hip_bicgstab_and_gmres_on_a_tetrahedral_mesh():
    from stormruler_amd import api

    pos, bf, lab, cells = _box(10)
    m = host_mesh.HostMesh.from_simplices(pos, bf, lab, cells)
    m.order_cells("morton")
    g = m.face_graph()
    ctx = api.Context(0)
    mat = m.create_operator(ctx)
    b = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    ref_op = oracle.StencilOperator(g, -1.0, 0.0)
    for cls, kind, tol in ((api.BiCgStabSolver, "bicgstab", 0.2), (api.GmresSolver, "gmres", 0.05)):
        x = api.DeviceVector(ctx, g.n_cells)
        s = cls()
        assert s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
        ref = oracle.solve(kind, ref_op, np.ones(g.n_cells))
        assert ref.converged and abs(s.iteration - ref.iterations) <= max(2, int(tol * ref.iterations)), (kind, s.iteration, ref.iterations)
        assert np.linalg.norm(x.to_numpy() - ref.x) <= 5e-6 * np.linalg.norm(ref.x)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_c_driver_partitions_and_solves_without_python(ranks, tmp_path):
    """tests/c/abi_two_ranks.c: TetGen files -> storm_hip_mesh_read_tetgen -> storm_hip_partition_rcb ->
    storm_hip_mesh_partition -> storm_hip_ctx_comm_init_host over pipes -> CG; against the same solve on one rank (in
    the driver) and against the oracle (here)."""
    import json
    import subprocess

    exe = os.path.join(ROOT, "tests", "c", "abi_two_ranks")
    pos, bf, lab, cells = _box(8)
    io_tetgen.write_tetgen(str(tmp_path / "box.1"), pos, bf, lab, cells)
    p = subprocess.run([exe, str(tmp_path / "box.1"), "3", str(ranks)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    g = io_tetgen.face_graph_from_simplices(pos, bf, lab, cells)
    ref = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells))
    assert out["ranks"] == ranks and out["cells"] == g.n_cells and out["converged"] == 1
    assert 0 < out["halo_rank0"] < out["owned_rank0"] and out["nbrs_rank0"] >= 1
    assert abs(out["iterations"] - ref.iterations) <= max(2, int(0.02 * ref.iterations))
    assert out["solution_rel_diff"] <= 1e-8
    assert abs(out["x_norm2"] - np.linalg.norm(ref.x)) <= 1e-8 * np.linalg.norm(ref.x)


@pytest.mark.gpu
def test_c_driver_on_the_reference_2d_mesh(golden):
    import json
    import subprocess

    u = golden["baseline_md_probe"]["unstructured"]
    exe = os.path.join(ROOT, "tests", "c", "abi_two_ranks")
    p = subprocess.run([exe, os.path.join(ROOT, u["mesh"]), "2", "2"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["cells"] == u["n_cells"] and out["converged"] == 1 and out["solution_rel_diff"] <= 1e-8


@pytest.mark.gpu
def test_renumbering_conjugates_the_tetrahedral_operator_bit_for_bit(tmp_path):
    """A size-independent property (at 12.6 M cells: tools/tet_conjugation_check.py, profiles/r08g_tet_conjugation.json):
    663 552 tetrahedra through the library's writer and reader; renumbering the cells (Morton, Hilbert, a scramble)
    conjugates the operator by a permutation -- faces keep their order, so every row sums the same terms in the same order:
    y_ordered == y_file[order] to the last bit, and 10 CG iterations leave the same residual to rounding."""
    from stormruler_amd import api

    pos, bf, lab, cells = _box(48)
    host_mesh.write_tetgen(str(tmp_path / "box.1"), pos, bf, lab, cells)
    n = 6 * 48 ** 3
    x_file = np.sin(0.37 * np.arange(n))
    ctx = api.Context(0)
    ys, res = {}, {}
    for mode in ("file", "morton", "hilbert", "random"):
        hm = host_mesh.HostMesh.read_tetgen(str(tmp_path / "box.1."), 3)
        order = np.arange(n)
        if mode == "random":
            hm.permute_cells(mesh.random_permutation(n))
        elif mode != "file":
            assert hm.order_cells(mode) == mode
        if mode != "file":
            order = np.ctypeslib.as_array(hm.view().global_id, shape=(n,)).copy()
        mat = hm.create_operator(ctx)
        st = mat.stats()
        assert st["max_row_len"] == 4 and st["tail_nnz"] == 0 and st["value_dictionary_size"] == 0
        y = api.DeviceVector(ctx, n)
        mat.apply(-1.0, 0.0, api.DeviceVector.from_numpy(ctx, x_file[order]), y)
        back = np.empty(n)
        back[order] = y.to_numpy()
        ys[mode] = back
        b, xs = api.DeviceVector(ctx, n), api.DeviceVector(ctx, n)
        api.fill_with(b, 1.0)
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = 10, 0.0, 0.0
        s.solve(xs, b, api.HipStencilOperator(mat, -1.0, 0.0))
        res[mode] = s.absolute_error
        mat.close()
        hm.close()
    for mode in ("morton", "hilbert", "random"):
        assert np.array_equal(ys[mode], ys["file"]), mode
        assert abs(res[mode] - res["file"]) <= 1e-12 * res["file"], mode
    ctx.close()
