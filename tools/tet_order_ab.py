#!/usr/bin/env python3
"""The tetrahedral box under different cell orderings: file order (cube-major, six cells per cube), the Z-order (Morton)
curve and the Hilbert curve of the cell centres (storm_hip_order_cells modes 1 / 3), and a seeded scramble.  Per ordering:
stand-alone SpMV (median of 60 launches on sin(0.37 i), rotating three (x, y) pairs) by SURVEY 8d's bytes, CG it/s."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stormruler_amd import api, host_mesh, io_tetgen, mesh  # noqa: E402

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pos, bf, cells = io_tetgen.tet_box(n3)
lab = np.ones(len(bf), np.int64)
ctx = api.Context(0)
for mode in ("file", "morton", "hilbert", "random"):
    hm = host_mesh.HostMesh.from_simplices(pos, bf, lab, cells)
    t0 = time.time()
    if mode == "random":
        hm.permute_cells(mesh.random_permutation(6 * n3 ** 3))
    elif mode != "file":
        hm.order_cells(mode)
    t_order = time.time() - t0
    mat = hm.create_operator(ctx)
    st = mat.stats()
    n = st["n_rows"]
    v = hm.view()
    band = np.abs(np.ctypeslib.as_array(v.inner, shape=(v.n_faces,)) - np.ctypeslib.as_array(v.outer, shape=(v.n_faces,)))
    xs = [api.DeviceVector.from_numpy(ctx, np.sin(0.37 * np.arange(n))) for _ in range(3)]
    ys = [api.DeviceVector(ctx, n) for _ in range(3)]
    for i in range(12):
        mat.apply(-1.0, 0.0, xs[i % 3], ys[i % 3])
    ctx.set_option("profile_spmv", 1)
    for i in range(60):
        mat.apply(-1.0, 0.0, xs[i % 3], ys[i % 3])
    smp = ctx.spmv_profile_samples()
    ctx.set_option("profile_spmv", 0)
    alg = 24 * n + 12 * st["nnz_offdiag"]
    b = api.DeviceVector(ctx, n)
    api.fill_with(b, 1.0)
    rates = []
    for _ in range(3):
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = 200, 0.0, 0.0
        x = api.DeviceVector(ctx, n)
        ctx.sync()
        t0 = time.perf_counter()
        s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
        ctx.sync()
        rates.append(200 / (time.perf_counter() - t0))
    print(json.dumps({"ordering": mode, "rows": n, "order_seconds": t_order, "spmv_median_ms": float(np.median(smp)),
                      "frac_8d": alg / (float(np.median(smp)) * 1e-3) / 1e9 / 8000.0, "cg_it_per_s": max(rates),
                      "cg_residual": s.absolute_error, "column_distance_median": int(np.median(band)),
                      "column_distance_p90": int(np.percentile(band, 90)), "column_distance_p99": int(np.percentile(band, 99))}), flush=True)
    del xs, ys, b, x
    mat.close()
    hm.close()
