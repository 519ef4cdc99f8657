"""Which record format and which solver path each BASELINE.json config gets -- asserted from `storm_hip_op_stats` and the
context's path counters, so that a refactor of the dispatch (csrc/spmv.hip's format ladder, csrc/solvers.hip's choice of
resident / latency / fused / kernel-per-statement loops) cannot silently move a config off the kernels its numbers in
bench.py and profiles/ were measured on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from stormruler_amd import api, cavity, mesh

    ctx = api.Context(0)
    yield api, mesh, cavity, ctx
    ctx.close()


def _counters(ctx):
    return {k: ctx.counter(k) for k in ("resident_solves", "latency_solves", "throughput_solves", "cg_fused_steps")}


def _solve(api, ctx, cls, mat, alpha, n, iters, **knobs):
    s = cls()
    s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
    for k, v in knobs.items():
        setattr(s, k, v)
    b, x = api.DeviceVector(ctx, n), api.DeviceVector(ctx, n)
    api.fill_with(b, 1.0)
    before = _counters(ctx)
    s.solve(x, b, api.HipStencilOperator(mat, alpha, 0.0))
    after = _counters(ctx)
    assert s.path_fallback == 0 and np.isfinite(s.absolute_error)
    return {k: after[k] - before[k] for k in after}


def test_config1_cg_64_runs_on_the_resident_path_on_canonical_records(env):
    api, mesh, cavity, ctx = env
    g = mesh.structured_box(64)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    st = mat.stats()
    assert st["paired_rows"] == 2 and 0 < st["value_dictionary_size"] <= 32 and st["tail_rows"] == 0  # format 4
    d = _solve(api, ctx, api.CgSolver, mat, -1.0, g.n_cells, 20)
    assert d == {"resident_solves": 1, "latency_solves": 0, "throughput_solves": 0, "cg_fused_steps": 0}
    mat.close()


def test_configs_2_and_3_256_cubed_run_the_lattice_kernels_and_the_fused_cg_step(env):
    api, mesh, cavity, ctx = env
    g = mesh.structured_box(256)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    st = mat.stats()
    # format 4 (canonical paired rows: 8 B/row) on the tiled lattice kernel
    assert st["paired_rows"] == 2 and st["tiled_planes"] == 2 and st["record_bytes"] == 8 * g.n_cells
    d = _solve(api, ctx, api.CgSolver, mat, -1.0, g.n_cells, 12)  # config 2: the marching CG step kernel
    assert d == {"resident_solves": 0, "latency_solves": 0, "throughput_solves": 1, "cg_fused_steps": 1}
    d = _solve(api, ctx, api.BiCgStabSolver, mat, -1.0, g.n_cells, 6)  # config 3's per-GPU block: the fused BiCGStab loop
    assert d == {"resident_solves": 0, "latency_solves": 0, "throughput_solves": 1, "cg_fused_steps": 0}
    mat.close()


def test_config4_gmres_convdiff_128_keeps_canonical_records_and_the_tiled_apply(env):
    api, mesh, cavity, ctx = env
    g = mesh.structured_box(128)
    wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    st = mat.stats()
    assert st["paired_rows"] == 2 and st["tiled_planes"] == 2 and st["tail_rows"] == 0
    d = _solve(api, ctx, api.GmresSolver, mat, 1.0, g.n_cells, 31, num_inner_iterations=30)
    assert d["resident_solves"] == 0 and d["latency_solves"] == 0  # (GMRES: the fused loop with the cooperative chain)
    mat.close()


def test_config5_cavity_pressure_solves_run_on_the_resident_path(env):
    api, mesh, cavity, ctx = env
    dev = cavity.CavityProjection(ctx, 128, nu=0.01)
    st = dev.A_p.matrix.stats() if hasattr(dev.A_p, "matrix") else None
    before = _counters(ctx)
    for _ in range(2):
        _, _, ok = dev.step()
        assert ok
    after = _counters(ctx)
    assert after["resident_solves"] - before["resident_solves"] == 2
    assert after["throughput_solves"] == before["throughput_solves"] and after["latency_solves"] == before["latency_solves"]
    if st is not None:
        assert st["paired_rows"] == 2


def test_general_meshes_keep_the_fp64_records_and_the_sell_kernel(env):
    """A jittered geometry (no two weights equal) and a renumbered box: what a Triangle / TetGen mesh gets."""
    api, mesh, cavity, ctx = env
    g = mesh.structured_box(48)
    gj = mesh.jitter_geometry(g, 1.0 / 48)
    mat = api.StencilMatrix.from_face_graph(ctx, gj)
    st = mat.stats()
    assert st["paired_rows"] == 0 and st["value_dictionary_size"] == 0 and st["record_bytes"] >= 12 * st["nnz_offdiag"]
    d = _solve(api, ctx, api.CgSolver, mat, -1.0, g.n_cells, 10)
    assert d["resident_solves"] == 0 and d["latency_solves"] == 1  # (110 592 rows: the latency path takes any small operator)
    mat.close()
    gp = mesh.permute_cells(g, mesh.random_permutation(g.n_cells))
    mat = api.StencilMatrix.from_face_graph(ctx, gp)
    st = mat.stats()
    assert st["paired_rows"] == 0 and st["value_dictionary_size"] > 0  # byte-indexed weights, int32 columns
    mat.close()
