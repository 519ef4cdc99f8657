#!/usr/bin/env python3
"""One 60-iteration CG solve of the 256^3 Poisson problem after a warm-up solve (for a kernel trace)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
solver = sys.argv[2] if len(sys.argv) > 2 else "cg"
ctx = api.Context(0)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
g = mesh.structured_box(n)
mat = api.StencilMatrix.from_face_graph(ctx, g)
b = api.DeviceVector(ctx, g.n_cells)
api.fill_with(b, 1.0)
for iters in [int(v) for v in os.environ.get("TRACE_ITERS", "100,60").split(",")]:
    s = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "gmres": api.GmresSolver}[solver]()
    s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
    x = api.DeviceVector(ctx, g.n_cells)
    s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
    ctx.sync()
