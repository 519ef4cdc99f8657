"""The library's cell ordering from geometry (csrc/ordering.hip, `storm_hip_order_cells`): host code in the role METIS
plays in the north star (the reference's hook: UnstructuredMesh::permute, Mallard/MeshUnstructured.hpp:443-459).  No GPU."""
import numpy as np
import pytest

from stormruler_amd import mesh


@pytest.mark.parametrize("shape", [(7, 5, 3), (24, 22, 27), (64, 48, 20), (128, 3, 2)])
def test_a_renumbered_box_gets_its_natural_order_back(shape):
    g = mesh.structured_box(*shape)
    perm = mesh.random_permutation(g.n_cells, seed=7)
    gs = mesh.permute_cells(g, perm)
    order, kind = mesh.geometric_ordering(gs)
    assert kind == "lattice" and np.array_equal(perm[order], np.arange(g.n_cells))
    gr = mesh.permute_cells(gs, order)
    assert np.array_equal(gr.inner, g.inner) and np.array_equal(gr.outer, g.outer) and np.array_equal(gr.center, g.center)


def test_a_2d_lattice_and_a_non_lattice():
    # a 2-D tensor grid
    xs, ys = np.meshgrid(np.linspace(0.0, 1.0, 31), np.linspace(-2.0, 3.0, 17), indexing="xy")
    c = np.stack([xs.ravel(), ys.ravel()], axis=1)
    perm = np.random.default_rng(3).permutation(c.shape[0])

    class G:
        pass

    g = G()
    g.n_cells, g.dim, g.center = c.shape[0], 2, c[perm]
    order, kind = mesh.geometric_ordering(g)
    assert kind == "lattice" and np.array_equal(perm[order], np.arange(c.shape[0]))
    with pytest.raises(Exception):
        g.center = np.random.default_rng(4).random((500, 2))
        g.n_cells = 500
        mesh.geometric_ordering(g, "lattice")


def test_morton_order_is_a_permutation_that_keeps_neighbours_close():
    g = mesh.structured_box(32)
    n = g.n_cells
    gj = mesh.permute_cells(mesh.jitter_geometry(g, 1.0 / 32), mesh.random_permutation(n, seed=11))
    order, kind = mesh.geometric_ordering(gj, "morton")
    assert kind == "morton" and np.array_equal(np.sort(order), np.arange(n))
    gm = mesh.permute_cells(gj, order)
    # a wavefront's 64 consecutive rows are a compact block: most of their neighbours lie within a few hundred rows
    near = np.abs(gm.inner - gm.outer) < 512
    assert near.mean() > 0.6
    # ... where the scramble alone has none
    assert (np.abs(gj.inner - gj.outer) < 512).mean() < 0.1
    # the Triangle mesh of the reference (2-D, no lattice): the automatic mode takes the curve
    import os

    from stormruler_amd import io_tetgen

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mesh", "square_nb.1.")
    t = io_tetgen.read_triangle(root)
    order, kind = mesh.geometric_ordering(t)
    assert kind == "morton" and np.array_equal(np.sort(order), np.arange(t.n_cells))


@pytest.mark.parametrize("dim", [2, 3])
def test_hilbert_order_visits_face_adjacent_cells(dim):
    """Mode 3: the Hilbert curve of the cell centres (Skilling's transposed index).  On a 2^k grid consecutive cells of the
    curve are always face neighbours -- no jump anywhere, where the Z-order curve jumps across the domain."""
    n = 16
    g = mesh.structured_box(n, n, n if dim == 3 else 1)
    if dim == 2:
        class G:
            pass

        g2 = G()
        g2.n_cells, g2.dim, g2.center = g.n_cells, 2, np.ascontiguousarray(g.center[:, :2])
        g = g2
    scr = np.random.default_rng(5).permutation(g.n_cells)
    c = np.ascontiguousarray(g.center[scr])

    class S:
        pass

    s = S()
    s.n_cells, s.dim, s.center = g.n_cells, dim, c
    order, kind = mesh.geometric_ordering(s, "hilbert")
    assert kind == "hilbert" and np.array_equal(np.sort(order), np.arange(g.n_cells))
    steps = np.abs(np.diff(c[order], axis=0)).sum(axis=1) * n
    assert np.allclose(steps, 1.0)
    order_m, kind_m = mesh.geometric_ordering(s, "morton")
    assert kind_m == "morton" and (np.abs(np.diff(c[order_m], axis=0)).sum(axis=1) * n).max() > 3.0


def test_non_finite_centres_are_rejected_not_indexed():
    class S:
        pass

    s = S()
    s.n_cells, s.dim = 1000, 3
    s.center = np.random.default_rng(1).random((1000, 3))
    s.center[417, 1] = np.nan
    for mode in ("auto", "morton", "hilbert"):
        with pytest.raises(Exception, match="non-finite"):
            mesh.geometric_ordering(s, mode)
