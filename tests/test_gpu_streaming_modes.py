"""The streaming kernels' access mode and the Gram-Schmidt pass width are performance choices only.

* blas1_nt (csrc/blas1_device.hpp: nt_dispatch): non-temporal or plain loads and stores of the BLAS-1 / solver kernels --
  the same values, the same order of every sum: solver histories and BLAS-1 results are BITWISE equal in both modes;
  the default mode switches at 6 * 2^20 rows per vector.
* mgs_steps (csrc/solvers.hip: mgs_pair_kernel / mgs_multi_kernel): two, three or four modified-Gram-Schmidt steps per
  pass over w (SolverGmres.hpp:157-161) -- the coefficients follow from bilinearity, so the histories agree to rounding,
  not bitwise.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from stormruler_amd import api, mesh

    ctx = api.Context(0)
    ctx.set_option("resident_path", 0)
    ctx.set_option("latency_path", 0)  # the kernel-per-statement loops: the kernels in question
    yield api, mesh, ctx
    ctx.close()


def _history(api, ctx, cls, mat, b_host, iters, **knobs):
    s = cls()
    s.record_history = True
    s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
    for k, v in knobs.items():
        setattr(s, k, v)
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, b_host.size)
    s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
    assert s.path_fallback == 0
    return np.asarray(s.history), x.to_numpy()


@pytest.mark.parametrize("shape", [(40, 36, 30), (33, 17, 9)])
def test_nontemporal_and_plain_streaming_give_bitwise_equal_solves(env, shape):
    api, mesh, ctx = env
    g = mesh.structured_box(*shape)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    b_host = 1.0 + 0.5 * np.cos(0.37 * np.arange(g.n_cells))
    for cls, iters, knobs in ((api.CgSolver, 40, {}), (api.BiCgStabSolver, 25, {}),
                              (api.GmresSolver, 45, {"num_inner_iterations": 20})):
        runs = []
        for mode in (0, 2):
            ctx.set_option("blas1_nt", mode)
            runs.append(_history(api, ctx, cls, mat, b_host, iters, **knobs))
        ctx.set_option("blas1_nt", 1)
        assert np.array_equal(runs[0][0], runs[1][0]), cls.__name__
        assert np.array_equal(runs[0][1], runs[1][1]), cls.__name__
    mat.close()


def test_nontemporal_mode_of_the_blas1_entry_points_is_bitwise_equal(env):
    api, mesh, ctx = env
    n = 100_003
    rng = np.random.default_rng(7)
    xs = [rng.standard_normal(n) for _ in range(9)]
    out = []
    for mode in (0, 2):
        ctx.set_option("blas1_nt", mode)
        v = [api.DeviceVector.from_numpy(ctx, x) for x in xs]
        y = api.DeviceVector.from_numpy(ctx, xs[0])
        y += 0.25 * v[1]    # axpy
        y <<= y - v[2]      # three streams
        y *= 1.5            # scale in place
        api.multi_axpy(y, [0.5 - 0.1 * j for j in range(8)], v[1:9])
        dots = api.multi_dot(y, v[1:9])
        out.append((y.to_numpy(), np.asarray(dots), api.norm_2(y), api.dot_product(y, v[3])))
    ctx.set_option("blas1_nt", 1)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2] and out[0][3] == out[1][3]


@pytest.mark.parametrize("shape", [(48, 40, 36), (31, 29, 23)])
def test_gram_schmidt_pass_widths_agree_to_rounding(env, shape):
    from stormruler_amd import mesh as mesh_mod
    api, mesh, ctx = env
    g = mesh.structured_box(*shape)
    nu, vel = 1e-2, (1.0, 0.5, 0.25)
    wi, wo, de = mesh_mod.convection_diffusion_weights(g, nu, vel)
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    b_host = np.sin(0.01 * np.arange(g.n_cells)) + 1.0
    ctx.set_option("coop_mgs", 0)  # the throughput-path passes, not the cooperative chain
    runs = {}
    for steps in (2, 3, 4):
        ctx.set_option("mgs_steps", steps)
        runs[steps] = _history(api, ctx, api.GmresSolver, mat, b_host, 64, num_inner_iterations=30)
    ctx.set_option("mgs_steps", 4)
    ctx.set_option("coop_mgs", 1)
    for steps in (3, 4):
        h2, hs = runs[2][0], runs[steps][0]
        assert hs.shape == h2.shape and not np.array_equal(hs, h2)  # (bitwise equal would mean the wider pass never ran)
        assert np.max(np.abs(hs / h2 - 1.0)) < 1e-8, steps
        assert np.linalg.norm(runs[steps][1] - runs[2][1]) <= 1e-9 * np.linalg.norm(runs[2][1]), steps
    mat.close()


def test_vector_arenas_change_where_vectors_lie_and_nothing_else():
    """Option vec_arena (context.hip): vectors of one size are slots of one allocation.  Same solves bit for bit; more
    vectors than an arena has slots, the pool trimmed under them, release in any order: all fine."""
    from stormruler_amd import api, mesh

    g = mesh.structured_box(64)  # 2 MiB vectors: arena-backed
    b_host = 1.0 + 0.25 * np.sin(0.05 * np.arange(g.n_cells))
    runs = []
    for arena in (0, 1):
        ctx = api.Context(0)
        ctx.set_option("vec_arena", arena)
        ctx.set_option("resident_path", 0)
        ctx.set_option("latency_path", 0)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        for cls, iters, knobs in ((api.CgSolver, 40, {}), (api.GmresSolver, 70, {"num_inner_iterations": 30})):
            runs.append(_history(api, ctx, cls, mat, b_host, iters, **knobs))
        vs = [api.DeviceVector.from_numpy(ctx, b_host + k) for k in range(21)]  # 8 + 16 slots: two arenas and a third begun
        ctx.set_option("pool_bytes", 0)  # trim: nothing of an arena may be handed back to the driver
        for k in (20, 0, 7, 13):
            assert np.array_equal(vs[k].to_numpy(), b_host + k)
        del vs[3:17]
        ws = [api.DeviceVector.from_numpy(ctx, b_host - k) for k in range(9)]  # released slots come back
        assert all(np.array_equal(w.to_numpy(), b_host - k) for k, w in enumerate(ws))
        assert np.array_equal(vs[-1].to_numpy(), b_host + 20)
        del ws, vs
        mat.close()
        ctx.close()
    for (h0, x0), (h1, x1) in zip(runs[:2], runs[2:]):
        assert np.array_equal(h0, h1) and np.array_equal(x0, x1)



