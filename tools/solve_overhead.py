#!/usr/bin/env python3
"""Fixed cost of one solve() call: wall time of K-iteration CG solves (tolerances off) for several K."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = mesh.structured_box(n)
ctx = api.Context(0)
mat = api.StencilMatrix.from_face_graph(ctx, g)
op = api.HipStencilOperator(mat, -1.0, 0.0)
b = api.DeviceVector(ctx, g.n_cells)
api.fill_with(b, 1.0)
out = {}
for rep in range(3):
    for K in (0, 1, 10, 100, 200, 400, 1000):
        x = api.DeviceVector(ctx, g.n_cells)
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = K, 0.0, 0.0
        ctx.sync()
        t0 = time.perf_counter()
        s.solve(x, b, op)
        ctx.sync()
        out.setdefault(K, []).append((time.perf_counter() - t0) * 1e3)
res = {K: min(v) for K, v in out.items()}
slope = (res[1000] - res[200]) / 800
print(json.dumps({"ms": res, "ms_per_iter_slope": slope, "intercept_ms": res[200] - 200 * slope}))
