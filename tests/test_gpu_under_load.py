"""The relaxed-atomic protocols under memory-system load: a second stream copies 1 GiB
buffers back to back while

  * the latency path (one cooperative kernel per solve; rows published for the other blocks' gathers, all-reduces
    that are their own barriers) and
  * the fused 256^3 loops (reductions finished inside the kernels by two levels of tickets)

run the same solve again and again: every residual history must be BITWISE the one of the idle device.  Plus option
`ticket_verify`: every k-th iteration the ticketed reductions are recomputed by the two-launch path and compared on
the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    a = torch.empty(1 << 27, dtype=torch.float64, device="cuda")  # 1 GiB
    b = torch.empty_like(a)
    a.fill_(1.0)
    yield api, mesh, ctx, torch, (a, b, torch.cuda.Stream())
    ctx.close()


def _history(api, ctx, cls, op, bh, n, iters):
    s = cls()
    s.record_history, s.num_iterations = True, iters
    s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
    x = api.DeviceVector(ctx, n)
    s.solve(x, bh, op)
    return np.array(s.history), x.to_numpy()


def _under_load(torch, load, run, solves):
    a, b, side = load
    out = []
    for _ in range(solves):
        with torch.cuda.stream(side):
            for _ in range(12):
                b.copy_(a, non_blocking=True)  # ~0.35 ms each: the device is never idle while the solve runs
        out.append(run())
    side.synchronize()
    return out


@pytest.mark.parametrize("publish", [1, 0], ids=["exchange", "store"])
@pytest.mark.parametrize("kind", ["cg", "bicgstab"])
def test_latency_path_is_bitwise_reproducible_under_load(env, kind, publish):
    api, mesh, ctx, torch, load = env
    g = mesh.structured_box(64)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    bh = api.DeviceVector.from_numpy(ctx, 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells)))
    cls, iters = (api.CgSolver, 600) if kind == "cg" else (api.BiCgStabSolver, 100)
    ctx.set_option("test_disable", 0 if publish else 16)  # (16: rows published by write-through stores)
    ctx.set_option("latency_path", 2)  # (the latency path itself; the resident path has its own test below)
    try:
        ref_h, ref_x = _history(api, ctx, cls, op, bh, g.n_cells, iters)
        assert np.all(np.isfinite(ref_h))
        for h, x in _under_load(torch, load, lambda: _history(api, ctx, cls, op, bh, g.n_cells, iters), 12):
            assert np.array_equal(h, ref_h) and np.array_equal(x, ref_x)
    finally:
        ctx.set_option("test_disable", 0)
        ctx.set_option("latency_path", 1)
        mat.close()


@pytest.mark.parametrize("kind,edge", [("cg", 64), ("bicgstab", 64), ("cg", 128), ("bicgstab", 128)])
def test_resident_path_is_bitwise_reproducible_under_load(env, kind, edge):
    """csrc/resident.hip: the surfaces of the blocks' boxes travel as self-validating granules, the reductions as tagged
    slots -- nothing depends on the order in which stores become visible, so a second stream copying 1 GiB buffers
    must not change a bit."""
    api, mesh, ctx, torch, load = env
    g = mesh.structured_box(edge)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    bh = api.DeviceVector.from_numpy(ctx, 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells)))
    cls, iters = (api.CgSolver, 600) if kind == "cg" else (api.BiCgStabSolver, 100)
    try:
        before = ctx.counter("resident_solves")
        ref_h, ref_x = _history(api, ctx, cls, op, bh, g.n_cells, iters)
        assert ctx.counter("resident_solves") == before + 1 and np.all(np.isfinite(ref_h))
        for h, x in _under_load(torch, load, lambda: _history(api, ctx, cls, op, bh, g.n_cells, iters), 8):
            assert np.array_equal(h, ref_h) and np.array_equal(x, ref_x)
    finally:
        mat.close()


@pytest.mark.parametrize("kind", ["cg", "bicgstab"])
def test_ticketed_reductions_are_bitwise_reproducible_under_load(env, kind):
    api, mesh, ctx, torch, load = env
    g = mesh.structured_box(256)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    bh = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(bh, 1.0)
    cls, iters = (api.CgSolver, 300) if kind == "cg" else (api.BiCgStabSolver, 40)
    ref_h, _ = _history(api, ctx, cls, op, bh, g.n_cells, iters)
    for h, _ in _under_load(torch, load, lambda: _history(api, ctx, cls, op, bh, g.n_cells, iters), 4):
        assert np.array_equal(h, ref_h)
    mat.close()


@pytest.mark.parametrize("kind", ["cg", "bicgstab"])
def test_ticket_verify_passes_and_catches_a_lost_partial(env, kind):
    """`ticket_verify = k`: every k-th iteration the in-kernel reductions are recomputed by the two-launch path and
    compared on the device; the solve's bits do not change.  With a partial sum dropped from the recomputation (test
    hook: what a stale read of one block's partial would look like) the solve returns an error."""
    api, mesh, ctx, torch, load = env
    g = mesh.structured_box(128)  # (beyond the latency path: the throughput loops with tickets)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    bh = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(bh, 1.0)
    cls, iters = (api.CgSolver, 120) if kind == "cg" else (api.BiCgStabSolver, 60)
    ctx.set_option("resident_path", 0)  # (... and not the resident path)
    try:
        h0, x0 = _history(api, ctx, cls, op, bh, g.n_cells, iters)
        ctx.set_option("ticket_verify", 3)
        h1, x1 = _history(api, ctx, cls, op, bh, g.n_cells, iters)
        assert np.array_equal(h0, h1) and np.array_equal(x0, x1)
        ctx.set_option("ticket_verify_inject", 1)
        with pytest.raises(api._lib.StormHipError, match="ticket_verify"):
            _history(api, ctx, cls, op, bh, g.n_cells, iters)
    finally:
        ctx.set_option("ticket_verify_inject", 0)
        ctx.set_option("ticket_verify", 0)
        ctx.set_option("resident_path", 1)
        mat.close()
