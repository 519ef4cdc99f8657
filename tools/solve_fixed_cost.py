#!/usr/bin/env python3
"""Wall time of a CG solve of the 256^3 problem against its iteration count: the fixed cost of a solve (vectors,
state upload, init residual, result) is what K = 20 steps carry on top of 20 iterations."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = api.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
g = mesh.structured_box(n)
mat = api.StencilMatrix.from_face_graph(ctx, g)
op = api.HipStencilOperator(mat, -1.0, 0.0)
if os.environ.get("FIXED_COST_LAMBDA"):  # the operator as a lambda (the general engine, csrc/krylov.hip)
    op = api.make_operator(lambda y, x: mat.apply(-1.0, 0.0, x, y))
b = api.DeviceVector(ctx, g.n_cells)
api.fill_with(b, 1.0)


def run(iters):
    t0 = time.perf_counter()
    x = api.DeviceVector(ctx, g.n_cells)
    t1 = time.perf_counter()
    s = api.CgSolver()
    s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
    s.solve(x, b, op)
    t2 = time.perf_counter()
    ctx.sync()
    t3 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6


run(200)
for iters in (0, 1, 2, 5, 20, 100):
    r = np.median([run(iters) for _ in range(15)], axis=0)
    print(f"K={iters:4d}: vector {r[0]:7.1f} us  solve {r[1]:8.1f} us  sync {r[2]:6.1f} us  total {r.sum():8.1f} us", flush=True)
