#!/usr/bin/env python3
"""Interleaved A/B of one library option on the CG rate:  tools/opt_ab.py <key> <v1,v2,...> [n] [iters]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

key, values = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 400
g = mesh.structured_box(n)
ctx = api.Context(0)
mat = api.StencilMatrix.from_face_graph(ctx, g)
op = api.HipStencilOperator(mat, -1.0, 0.0)
b = api.DeviceVector(ctx, g.n_cells)
api.fill_with(b, 1.0)
res = {}
for rnd in range(4):
    for v in values:
        ctx.set_option(key, v)
        x = api.DeviceVector(ctx, g.n_cells)
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
        ctx.sync()
        t0 = time.perf_counter()
        s.solve(x, b, op)
        ctx.sync()
        res.setdefault(v, []).append(iters / (time.perf_counter() - t0))
print(json.dumps({str(k): [round(t, 1) for t in v] for k, v in res.items()}))
