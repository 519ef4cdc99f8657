#!/usr/bin/env python3
"""Size robustness: an n^3 Poisson problem far beyond the 256 MiB Infinity Cache (default 512^3:
134 M rows, 0.8 G off-diagonal entries, 10.7 GB of slice records, 1 GiB per vector)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t = time.time()
g = mesh.structured_box(n)
t_mesh = time.time() - t
ctx = api.Context(0)
t = time.time()
mat = api.StencilMatrix.from_face_graph(ctx, g)
t_op = time.time() - t
st = mat.stats()
N = g.n_cells
b_spmv = 24 * N + 12 * st["nnz_offdiag"]
x, y = api.DeviceVector(ctx, N), api.DeviceVector(ctx, N)
api.fill_with(x, 1.0)
mat.apply(-1.0, 0.0, x, y)
r = y.to_numpy().reshape(n, n, n)
ok_const = bool(np.all(r[1:-1, 1:-1, 1:-1] == 0.0) and np.isclose(r[0, 0, 0], 6.0 * n * n))
for _ in range(3):
    mat.apply(-1.0, 0.0, x, y)
ctx.timer_start()
for _ in range(10):
    mat.apply(-1.0, 0.0, x, y)
ms = ctx.timer_stop() / 10
b = api.DeviceVector(ctx, N)
api.fill_with(b, 1.0)
xs = api.DeviceVector(ctx, N)
s = api.CgSolver()
s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = 60, 0.0, 0.0
s.record_history = True
ctx.sync()
t = time.time()
s.solve(xs, b, api.HipStencilOperator(mat, -1.0, 0.0))
ctx.sync()
t_cg = time.time() - t
true_res = api.HipStencilOperator(mat, -1.0, 0.0).ResidualNorm(b, xs)
print(json.dumps({"n": n, "rows": N, "nnz_offdiag": st["nnz_offdiag"], "operator_device_GB": st["device_bytes"] / 1e9,
                  "mesh_seconds": t_mesh, "operator_build_seconds": t_op, "constant_vector_check": ok_const,
                  "spmv_ms": ms, "spmv_algorithmic_GBs": b_spmv / ms / 1e6, "spmv_frac_of_8TBs": b_spmv / ms / 1e6 / 8000,
                  "cg_iterations_per_sec": 60 / t_cg, "cg_ms_per_iteration": t_cg / 60 * 1e3,
                  "recurrence_vs_true_residual_rel_diff": abs(true_res - s.history[-1]) / s.history[0]}))
