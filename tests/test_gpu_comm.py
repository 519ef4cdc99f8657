"""The multi-rank code path on ONE GPU: an RCCL communicator of size 1 and a z-periodic mesh whose
halo is exchanged with the rank itself (a self-neighbour = periodic boundary).  Exercises the pack
kernel, grouped ncclSend/ncclRecv on the comm stream, the interior/boundary split of the SpMV, the
stream/event ordering and the all-reduce path of every solver -- everything the 8-GPU run uses
except a second device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _periodic_z_local_graph(nx, ny, nz):
    from stormruler_amd import mesh

    return mesh.periodic_z_local_graph(nx, ny, nz)


@pytest.fixture(scope="module")
def setup():
    from oracle import oracle
    from stormruler_amd import api

    ctx = api.Context(0)
    ctx.comm_init(api.Context.comm_unique_id(), 1, 0)
    loc, send_idx = _periodic_z_local_graph(20, 12, 9)
    mat = api.StencilMatrix.from_face_graph(ctx, loc)
    n2p = loc.n_halo
    mat.set_halo([0], [0, n2p], send_idx, [0, n2p])
    ref_op = oracle.StencilOperator(loc, -1.0, 0.05)

    def ref_apply(x_owned):
        xf = np.concatenate([x_owned, x_owned[send_idx]])
        return ref_op.apply(xf)[: loc.n_cells]

    yield api, ctx, loc, mat, ref_apply, oracle
    mat.close()
    ctx.close()


def test_halo_exchange_and_split_spmv(setup):
    api, ctx, loc, mat, ref_apply, _ = setup
    st = mat.stats()
    assert 0 < st["n_interior_slices"] < st["n_slices"]
    x = np.sin(0.37 * np.arange(loc.n_cells))
    xv = api.DeviceVector.from_numpy(ctx, x, n_halo=loc.n_halo)
    yv = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
    for _ in range(3):  # repeated: event / stream ordering must hold across applies
        mat.apply(-1.0, 0.05, xv, yv)
    y_ref = ref_apply(x)
    assert np.abs(yv.to_numpy() - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    # the halo tail of x now holds the periodic images
    assert np.array_equal(xv.to_numpy(with_halo=True)[loc.n_cells:], x[np.concatenate([np.arange(loc.n_cells - 240, loc.n_cells), np.arange(240)])])
    # reductions go through the size-1 all-reduce
    assert abs(api.dot_product(xv, yv) - float(x @ y_ref)) <= 1e-11 * abs(float(x @ y_ref))


def test_operator_with_plan_needs_a_communicator():
    from stormruler_amd import api

    c2 = api.Context(0)
    loc, send_idx = _periodic_z_local_graph(8, 8, 8)
    m2 = api.StencilMatrix.from_face_graph(c2, loc)
    m2.set_halo([0], [0, loc.n_halo], send_idx, [0, loc.n_halo])
    x = api.DeviceVector(c2, loc.n_cells, loc.n_halo)
    y = api.DeviceVector(c2, loc.n_cells, loc.n_halo)
    with pytest.raises(api._lib.StormHipError):
        m2.apply(1.0, 0.0, x, y)
    m2.close()
    c2.close()


@pytest.mark.parametrize("kind,tol", [("cg", 1e-8), ("bicgstab", 5e-6), ("gmres", 5e-6)])
def test_solvers_through_the_comm_path(setup, kind, tol):
    api, ctx, loc, mat, ref_apply, oracle = setup
    b_host = np.cos(0.05 * np.arange(loc.n_cells)) + 0.3
    cls = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "gmres": api.GmresSolver}[kind]
    s = cls()
    if kind == "gmres":
        s.num_inner_iterations = 20
    b = api.DeviceVector.from_numpy(ctx, b_host, n_halo=loc.n_halo)
    x = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
    assert s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.05))
    ref = oracle.solve(kind, oracle.CallbackOperator(loc.n_cells, ref_apply), b_host,
                       num_inner_iterations=20)
    assert ref.converged and abs(s.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations))
    assert np.linalg.norm(x.to_numpy() - ref.x) <= tol * np.linalg.norm(ref.x)


@pytest.mark.parametrize("kind", ["cg", "bicgstab"])
def test_rccl_comm_breakdown_on_the_size_one_communicator(setup, kind):
    """Option profile_comm: device timestamps around every step of the RCCL transport's halo exchange and all-reduces
    (csrc/comm.hip) -- the `comm_breakdown` of an N > 1 bench line on this transport (bench.py).  On the size-1
    communicator every key must be there, the counts must be those of the loop, and the times plausible."""
    api, ctx, loc, mat, ref_apply, oracle = setup
    K = 40
    b = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
    api.fill_with(b, 1.0)
    s = (api.CgSolver if kind == "cg" else api.BiCgStabSolver)()
    s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = K, 0.0, 0.0
    x = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
    s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.05))
    plain = x.to_numpy()
    ctx.set_option("profile_comm", 1)
    x2 = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
    s.solve(x2, b, api.HipStencilOperator(mat, -1.0, 0.05))
    bd = ctx.rccl_profile(K)
    ctx.set_option("profile_comm", 0)
    assert np.array_equal(x2.to_numpy(), plain)  # the stamps change no value
    for k in ("transport", "iterations_covered", "halo_exchanges_per_iteration", "allreduces_per_iteration",
              "event_to_comm_stream_us_each", "pack_us_each", "sendrecv_us_each", "halo_unhidden_wait_us_each",
              "halo_done_to_boundary_rows_us_each", "allreduce_us_each", "halo_unhidden_us_per_iteration",
              "allreduce_us_per_iteration", "note"):
        assert k in bd, k
    applies = 1 if kind == "cg" else 2
    assert abs(bd["halo_exchanges_per_iteration"] - applies) <= 3.0 / K  # (+ the initial residual's apply)
    # CG 1 + 1 all-reduces per iteration; BiCGStab's five reductions batched to 3 calls
    assert (1.9 if kind == "cg" else 2.9) <= bd["allreduces_per_iteration"] <= (2.2 if kind == "cg" else 3.2), bd
    for k in ("event_to_comm_stream_us_each", "pack_us_each", "sendrecv_us_each", "halo_done_to_boundary_rows_us_each", "allreduce_us_each"):
        assert 0.0 < bd[k] < 500.0, (k, bd[k])
    assert bd["halo_unhidden_wait_us_each"] >= 0.0
    with pytest.raises(api._lib.StormHipError):
        api.Context(0).counter("rccl_prof_exchanges")  # needs the transport and the option


@pytest.mark.parametrize("kind", ["cg", "bicgstab", "gmres"])
def test_flag_wait_and_event_wait_give_the_same_solve(setup, kind):
    """Option rccl_flag_wait (default 1): the boundary rows are released by a flag in device memory that a one-thread kernel
    sets behind the exchange on the comm stream, instead of a cross-stream event.  Only WHEN the boundary launch starts
    changes: the same bits either way."""
    api, ctx, loc, mat, ref_apply, oracle = setup
    b = api.DeviceVector.from_numpy(ctx, np.cos(0.05 * np.arange(loc.n_cells)) + 0.3, n_halo=loc.n_halo)
    cls = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "gmres": api.GmresSolver}[kind]
    out = {}
    for flag in (1, 0, 1):
        ctx.set_option("rccl_flag_wait", flag)
        s = cls()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = 25, 0.0, 0.0
        x = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
        s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.05))
        out.setdefault(flag, []).append((s.absolute_error, x.to_numpy()))
    ctx.set_option("rccl_flag_wait", 1)
    for err, xs in out[1][1:] + out[0]:
        assert err == out[1][0][0] and np.array_equal(xs, out[1][0][1])


@pytest.mark.parametrize("shape", [(20, 12, 9), (32, 32, 24)])
def test_the_plain_form_of_the_rccl_transport_against_the_oracle(shape):
    """bench.py's `rccl-plain` fallback = the RCCL transport with every refinement that has only ever run on a size-1
    communicator switched off AT ONCE (bench.RCCL_PLAIN_OPTIONS: cross-stream events instead of flag waits, two-launch
    reductions, no early halo, no fused step).  Each switch has its own test here; this one holds the combination the
    fallback actually runs -- CG, BiCGStab and GMRES against the oracle, on a box that gets the general records and on one
    whose interior planes get the lattice records."""
    import os
    import sys

    from oracle import oracle
    from stormruler_amd import api

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    ctx = api.Context(0)
    try:
        for key in bench.RCCL_PLAIN_OPTIONS:
            ctx.set_option(key, 0)
        ctx.comm_init(api.Context.comm_unique_id(), 1, 0)
        loc, send_idx = _periodic_z_local_graph(*shape)
        mat = api.StencilMatrix.from_face_graph(ctx, loc)
        mat.set_halo([0], [0, loc.n_halo], send_idx, [0, loc.n_halo])
        ref_op = oracle.StencilOperator(loc, -1.0, 0.05)
        ref_apply = lambda x: ref_op.apply(np.concatenate([x, x[send_idx]]))[: loc.n_cells]  # noqa: E731
        b_host = np.cos(0.05 * np.arange(loc.n_cells)) + 0.3
        for kind, tol in (("cg", 1e-8), ("bicgstab", 5e-6), ("gmres", 5e-6)):
            s = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "gmres": api.GmresSolver}[kind]()
            if kind == "gmres":
                s.num_inner_iterations = 20
            b = api.DeviceVector.from_numpy(ctx, b_host, n_halo=loc.n_halo)
            x = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
            assert s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.05)), kind
            ref = oracle.solve(kind, oracle.CallbackOperator(loc.n_cells, ref_apply), b_host, num_inner_iterations=20)
            assert ref.converged and abs(s.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations)), (kind, s.iteration, ref.iterations)
            assert np.linalg.norm(x.to_numpy() - ref.x) <= tol * np.linalg.norm(ref.x), kind
        mat.close()
    finally:
        ctx.close()


def test_gmres_cgs2_through_the_comm_path(setup):
    api, ctx, loc, mat, ref_apply, oracle = setup
    b_host = np.ones(loc.n_cells)
    s = api.GmresSolver()
    s.num_inner_iterations, s.gram_schmidt = 15, 1
    b = api.DeviceVector.from_numpy(ctx, b_host, n_halo=loc.n_halo)
    x = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
    assert s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.05))
    ref = oracle.solve("gmres", oracle.CallbackOperator(loc.n_cells, ref_apply), b_host, num_inner_iterations=15)
    assert abs(s.iteration - ref.iterations) <= max(2, int(0.05 * ref.iterations))
    assert np.linalg.norm(x.to_numpy() - ref.x) <= 5e-6 * np.linalg.norm(ref.x)


@pytest.mark.parametrize("shape", [(32, 32, 24), (64, 16, 40)])
def test_fused_cg_step_over_rccl_gives_the_unfused_iteration(shape):
    """Round 4: on the RCCL transport too the SpMV launch ends the previous CG iteration (x += alpha p, p' = r + beta p)
    -- the boundary planes of p' are formed and packed by a small kernel on the comm stream and travel while the
    marching launch runs the interior planes; the boundary launch waits for them.  Same iteration counts, histories to
    1e-9 and solutions as the kernel-per-statement loop on the same transport (option rccl_fused = 0), and as the
    oracle on the periodic operator."""
    from oracle import oracle
    from stormruler_amd import api

    loc, send_idx = _periodic_z_local_graph(*shape)
    ref_op = oracle.StencilOperator(loc, -1.0, 0.05)

    def ref_apply(x_owned):
        return ref_op.apply(np.concatenate([x_owned, x_owned[send_idx]]))[: loc.n_cells]

    b_host = np.cos(0.05 * np.arange(loc.n_cells)) + 0.3
    runs = {}
    for fused in (1, 2, 0):  # (2: fused, the local sums by partials + a final pass instead of tickets)
        ctx = api.Context(0)
        ctx.set_option("spmv_canon_tile_min_rows", 0)  # (the lattice kernels on a slab this small)
        ctx.set_option("rccl_fused", min(fused, 1))
        ctx.set_option("rccl_ticket", int(fused == 1))
        ctx.comm_init(api.Context.comm_unique_id(), 1, 0)
        mat = api.StencilMatrix.from_face_graph(ctx, loc)
        mat.set_halo([0], [0, loc.n_halo], send_idx, [0, loc.n_halo])
        s = api.CgSolver()
        s.record_history = True
        b = api.DeviceVector.from_numpy(ctx, b_host, n_halo=loc.n_halo)
        x = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
        before = ctx.counter("cg_fused_steps")
        assert s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.05))
        runs[fused] = (s.iteration, np.array(s.history), x.to_numpy(), ctx.counter("cg_fused_steps") - before)
        # ... and a solve stopped by the iteration limit (the x update of the last iteration runs behind the loop)
        s2 = api.CgSolver()
        s2.num_iterations, s2.absolute_error_tolerance, s2.relative_error_tolerance = 7, 0.0, 0.0
        x2 = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
        s2.solve(x2, b, api.HipStencilOperator(mat, -1.0, 0.05))
        runs[(fused, "limit")] = (s2.absolute_error, x2.to_numpy())
        mat.close()
        ctx.close()
    assert runs[1][3] == 1 and runs[2][3] == 1 and runs[0][3] == 0
    for f in (1, 2):
        assert runs[f][0] == runs[0][0]
        assert np.allclose(runs[f][1], runs[0][1], rtol=1e-9)
        assert np.linalg.norm(runs[f][2] - runs[0][2]) <= 1e-10 * np.linalg.norm(runs[0][2])
        assert np.isclose(runs[(f, "limit")][0], runs[(0, "limit")][0], rtol=1e-10)
        assert np.linalg.norm(runs[(f, "limit")][1] - runs[(0, "limit")][1]) <= 1e-10 * np.linalg.norm(runs[(0, "limit")][1])
    ref = oracle.solve("cg", oracle.CallbackOperator(loc.n_cells, ref_apply), b_host)
    assert abs(runs[1][0] - ref.iterations) <= max(2, int(0.02 * ref.iterations))
    assert np.linalg.norm(runs[1][2] - ref.x) <= 1e-8 * np.linalg.norm(ref.x)


@pytest.mark.parametrize("kind", ["bicgstab", "cg"])
@pytest.mark.parametrize("shape", [(32, 32, 24), (64, 16, 40)])
def test_bicgstab_over_rccl_sends_the_halo_of_s_and_p_before_the_update_kernels(shape, kind):
    """Round 4: on the RCCL transport the halo of s = r - alpha v and of p' = r + beta (p - omega v) leaves BEFORE the kernel that
    forms the vector on the owned rows runs -- the rows to send are formed by a small kernel with the owner's expression from
    the operands as they are, and travel under the update and the interior rows of the apply (option rccl_early_halo).
    Bitwise the exchange begun by the apply itself, and the oracle's solve on the periodic operator.  kind = "cg": the
    kernel-per-statement CG loop (rccl_fused = 0: what a general operator gets) sends the halo of p' = r + beta p before
    cg_xp_kernel forms it."""
    from oracle import oracle
    from stormruler_amd import api

    loc, send_idx = _periodic_z_local_graph(*shape)
    ref_op = oracle.StencilOperator(loc, -1.0, 0.05)

    def ref_apply(x_owned):
        return ref_op.apply(np.concatenate([x_owned, x_owned[send_idx]]))[: loc.n_cells]

    b_host = np.cos(0.05 * np.arange(loc.n_cells)) + 0.3
    runs = {}
    for early in (1, 0):
        ctx = api.Context(0)
        ctx.set_option("spmv_canon_tile_min_rows", 0)
        ctx.set_option("rccl_early_halo", early)
        ctx.set_option("rccl_fused", 0)
        # every iteration's ticketed sums are recomputed and compared on the device, AND (BiCGStab) the alpha the early halo
        # of s was formed with must be, to the bit, the update kernel's: a mismatch fails the solve
        ctx.set_option("ticket_verify", 1)
        ctx.comm_init(api.Context.comm_unique_id(), 1, 0)
        mat = api.StencilMatrix.from_face_graph(ctx, loc)
        mat.set_halo([0], [0, loc.n_halo], send_idx, [0, loc.n_halo])
        s = api.BiCgStabSolver() if kind == "bicgstab" else api.CgSolver()
        s.record_history = True
        b = api.DeviceVector.from_numpy(ctx, b_host, n_halo=loc.n_halo)
        x = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
        assert s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.05))
        runs[early] = (s.iteration, np.array(s.history), x.to_numpy())
        mat.close()
        ctx.close()
    assert runs[1][0] == runs[0][0]
    assert np.array_equal(runs[1][1], runs[0][1]) and np.array_equal(runs[1][2], runs[0][2])
    ref = oracle.solve(kind, oracle.CallbackOperator(loc.n_cells, ref_apply), b_host)
    # (BiCGStab's iteration count is a draw among roundings -- NOTES.md 5c --: the solution is what is compared)
    assert abs(runs[1][0] - ref.iterations) <= max(3, int(0.2 * ref.iterations))
    assert np.linalg.norm(runs[1][2] - ref.x) <= 1e-5 * np.linalg.norm(ref.x)
