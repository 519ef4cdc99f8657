"""Host logic of the row partition: slab generator == generic partitioner, halo plans pair up,
and a 2-way partitioned operator apply (oracle compute, numpy halo copy) equals the global one."""
import numpy as np
import pytest

from oracle import oracle
from stormruler_amd import mesh, partition


def _lengths(nx, ny, nzg):
    return (1.0, ny / nx, nzg / nx)


@pytest.mark.parametrize("n_ranks", [2, 3, 4])
def test_slab_generator_matches_generic_partitioner(n_ranks):
    nx, ny, nzl = 6, 5, 3
    nzg = nzl * n_ranks
    g = mesh.structured_box(nx, ny, nzg, lengths=_lengths(nx, ny, nzg))
    part = (np.arange(g.n_cells) // (nx * ny)) // nzl
    for r in range(n_ranks):
        a, plan = partition.slab_partition(nx, ny, nzl, n_ranks, r)
        b = partition.partition_graph(g, part, r)
        a.validate(), b.validate()
        assert a.n_cells == b.n_cells and a.n_halo == b.n_halo
        assert np.array_equal(a.global_id, b.global_id) and np.array_equal(a.halo_owner, b.halo_owner)
        assert np.array_equal(a.inner, b.inner) and np.array_equal(a.outer, b.outer)
        assert np.allclose(a.area, b.area) and np.allclose(a.center, b.center) and np.allclose(a.volume, b.volume)
        assert np.array_equal(a.b_cell, b.b_cell) and np.allclose(a.b_center, b.b_center)
        plan_b = partition.halo_plan(b, r)
        for f in ("nbr_rank", "send_ptr", "send_idx", "recv_ptr"):
            assert np.array_equal(getattr(plan, f), getattr(plan_b, f))


def test_halo_plans_pair_up_and_apply_matches_global():
    rng = np.random.default_rng(0)
    g = mesh.structured_box(7, 6, 8)
    # an irregular 3-way partition (not slabs): exercises multi-neighbour plans
    part = (rng.random(g.n_cells) * 3).astype(np.int64)
    locs = [partition.partition_graph(g, part, r) for r in range(3)]
    plans = [partition.halo_plan(l, r) for r, l in enumerate(locs)]
    x = rng.standard_normal(g.n_cells)
    y_glob = oracle.StencilOperator(g, -1.0, 0.3).apply(x)
    # what rank r sends to q must be, in order, what q expects from r
    for r in range(3):
        pr, lr = plans[r], locs[r]
        for qi, q in enumerate(pr.nbr_rank):
            sent_gids = lr.global_id[pr.send_idx[pr.send_ptr[qi]:pr.send_ptr[qi + 1]]]
            pq, lq = plans[q], locs[q]
            j = list(pq.nbr_rank).index(r)
            want = lq.global_id[lq.n_cells + pq.recv_ptr[j]: lq.n_cells + pq.recv_ptr[j + 1]]
            assert np.array_equal(sent_gids, want)
    # emulate the exchange and apply locally
    y = np.empty_like(x)
    for r in range(3):
        lr = locs[r]
        xl = x[lr.global_id]  # owned + halo values (halo filled as the exchange would)
        yl = oracle.StencilOperator(lr, -1.0, 0.3).apply(xl)
        y[lr.global_id[: lr.n_cells]] = yl[: lr.n_cells]
    assert np.abs(y - y_glob).max() <= 1e-12 * np.abs(y_glob).max()


def test_orderings_are_permutations():
    g = mesh.structured_box(8, 6, 4)
    for perm in (mesh.random_permutation(g.n_cells), mesh.tile_ordering(8, 6, 4, 2, 2), mesh.rcm_ordering(g)):
        assert np.array_equal(np.sort(perm), np.arange(g.n_cells))
        gp = mesh.permute_cells(g, perm)
        gp.validate()
        x = np.sin(np.arange(g.n_cells) * 0.3)
        y = oracle.StencilOperator(g, -1.0, 0.0).apply(x)
        yp = oracle.StencilOperator(gp, -1.0, 0.0).apply(x[perm])
        assert np.abs(yp - y[perm]).max() <= 1e-12 * np.abs(y).max()
    # RCM actually reduces the bandwidth of a scrambled mesh
    gs = mesh.permute_cells(g, mesh.random_permutation(g.n_cells))
    bw0 = np.abs(gs.inner - gs.outer).max()
    gr = mesh.permute_cells(gs, mesh.rcm_ordering(gs))
    assert np.abs(gr.inner - gr.outer).max() < bw0 / 2


@pytest.mark.parametrize("n_parts", [2, 3, 5, 8])
def test_rcb_partition_on_the_unstructured_reference_mesh(n_parts):
    """General meshes: recursive coordinate bisection + the generic partitioner on the reference's own
    Triangle mesh; balanced parts, pairing halo plans, partitioned apply == global apply."""
    import os

    from stormruler_amd import io_tetgen

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = io_tetgen.read_triangle(os.path.join(root, "tests", "golden", "mesh", "square_nb.1."))
    part = partition.rcb_partition(g.center, n_parts)
    sizes = np.bincount(part, minlength=n_parts)
    assert sizes.sum() == g.n_cells and sizes.max() - sizes.min() <= n_parts
    x = np.sin(3 * g.center[:, 0]) * np.cos(7 * g.center[:, 1])
    y_glob = oracle.StencilOperator(g, -1e-2, 1.0).apply(x)
    y = np.empty_like(x)
    cut = 0
    for r in range(n_parts):
        loc = partition.partition_graph(g, part, r)
        loc.validate()
        plan = partition.halo_plan(loc, r)
        assert plan.recv_ptr[-1] == loc.n_halo and np.all(plan.nbr_rank != r)
        cut += loc.n_halo
        yl = oracle.StencilOperator(loc, -1e-2, 1.0).apply(x[loc.global_id])
        y[loc.global_id[: loc.n_cells]] = yl[: loc.n_cells]
    assert np.abs(y - y_glob).max() <= 1e-13 * np.abs(y_glob).max()
    assert cut < 0.25 * g.n_cells  # geometric cuts keep the halo small


def _mesh_for(kind):
    import os

    from stormruler_amd import io_tetgen

    if kind == "triangles":
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        return io_tetgen.read_triangle(os.path.join(root, "tests", "golden", "mesh", "square_nb.1."))
    if kind == "tetrahedra":
        pos, bf, cells = io_tetgen.tet_box(6)
        return io_tetgen.face_graph_from_simplices(pos, bf, np.ones(len(bf), np.int64), cells)
    return mesh.structured_box(9, 7, 12)


@pytest.mark.parametrize("kind", ["triangles", "tetrahedra", "box"])
@pytest.mark.parametrize("n_parts", [2, 3, 5, 8])
def test_native_partition_equals_the_numpy_restatement_array_for_array(kind, n_parts):
    """csrc/mesh_host.hip against stormruler_amd/partition.py: the cell -> rank map of the recursive coordinate bisection,
    every rank's local graph (owned cells, halo cells grouped by owner in ascending global id, faces, boundary faces,
    geometry) and every halo plan (neighbours, send lists, receive ranges) -- identical arrays."""
    from stormruler_amd import host_mesh

    g = _mesh_for(kind)
    part = partition.rcb_partition(g.center, n_parts)
    assert np.array_equal(host_mesh.partition_rcb(g.center, n_parts), part)
    hm = host_mesh.HostMesh.from_face_graph(g)
    for r in range(n_parts):
        want = partition.partition_graph(g, part, r)
        plan = partition.halo_plan(want, r)
        loc = hm.partition(part, n_parts, r)
        got, got_plan = loc.face_graph(), loc.halo_plan()
        assert (got.n_cells, got.n_halo, got.dim) == (want.n_cells, want.n_halo, want.dim)
        for name in ("inner", "outer", "area", "center", "volume", "b_cell", "b_area", "b_center", "global_id", "halo_owner"):
            a, b = np.asarray(getattr(got, name)), np.asarray(getattr(want, name))
            assert a.shape == b.shape and np.array_equal(a, b), (name, r)
        for name in ("nbr_rank", "send_ptr", "send_idx", "recv_ptr"):
            assert np.array_equal(np.asarray(getattr(got_plan, name)), np.asarray(getattr(plan, name))), (name, r)
        loc.close()
    hm.close()


def test_native_slabs_cut_a_box_into_its_planes():
    from stormruler_amd import host_mesh

    g = mesh.structured_box(6, 5, 12)
    for n_parts in (2, 3, 4):
        part = host_mesh.partition_slabs(g.center, 2, n_parts)
        want = (np.arange(g.n_cells) // (6 * 5)) // (12 // n_parts)
        assert np.array_equal(part, want)
