#!/usr/bin/env python3
"""256^3 CG (and BiCGStab) rate against the distance between the vectors of a context's arena (option vec_arena_skew_kib:
pitch = the vector rounded up to 2 MiB + skew), one fresh context per value; `off` = every vector its own allocation.
    python tools/arena_sweep.py [skews_KiB comma separated] [n]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

skews = sys.argv[1].split(",") if len(sys.argv) > 1 else ["off", "0", "256", "512", "1024", "1536", "2048", "3072", "4096", "6144", "8192", "12288"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = mesh.structured_box(n)
for sk in skews:
    ctx = api.Context(0)
    if sk == "off":
        ctx.set_option("vec_arena", 0)
    else:
        ctx.set_option("vec_arena_skew_kib", int(sk))
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    row = {"skew_KiB": sk}
    for name, cls, iters in (("cg", api.CgSolver, 300), ("bicgstab", api.BiCgStabSolver, 120)):
        rates = []
        for _ in range(4):
            x = api.DeviceVector(ctx, g.n_cells)
            s = cls()
            s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
            ctx.sync()
            t0 = time.perf_counter()
            s.solve(x, b, op)
            ctx.sync()
            rates.append(round(iters / (time.perf_counter() - t0), 1))
            del x
        row[name] = rates
    print(json.dumps(row), flush=True)
    mat.close()
    del b
    ctx.close()
