#!/usr/bin/env python3
"""Full-size, full-solve parity: 256^3 Poisson CG to the default tolerances on the GPU and on the CPU
oracle (minutes of CPU time; not part of the test suite).  Prints one JSON line for profiles/."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from stormruler_amd import api, mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = mesh.structured_box(n)
ctx = api.Context(0)
mat = api.StencilMatrix.from_face_graph(ctx, g)
b, x = api.DeviceVector(ctx, g.n_cells), api.DeviceVector(ctx, g.n_cells)
api.fill_with(b, 1.0)
s = api.CgSolver()
s.record_history = True
ctx.sync()
t = time.perf_counter()
ok = s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
ctx.sync()
tg = time.perf_counter() - t
t = time.perf_counter()
r = oracle.solve("cg", oracle.StencilOperator(g, -1.0, 0.0), np.ones(g.n_cells))
tc = time.perf_counter() - t
xg = x.to_numpy()
m = min(len(s.history), len(r.history))
print(json.dumps({
    "case": f"CG, {n}^3 Poisson, default tolerances", "gpu_converged": ok, "cpu_converged": r.converged,
    "gpu_iterations": s.iteration, "cpu_iterations": r.iterations,
    "solution_rel_l2_diff": float(np.linalg.norm(xg - r.x) / np.linalg.norm(r.x)),
    "max_rel_history_diff_first_100": float(np.max(np.abs(s.history[:min(m, 100)] - r.history[:min(m, 100)]) / r.history[:min(m, 100)])),
    "gpu_final_rel_residual": s.relative_error, "cpu_final_rel_residual": r.relative_error,
    "gpu_seconds": tg, "cpu_seconds_1_thread": tc, "speedup": tc / tg}))
