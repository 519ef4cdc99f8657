#!/usr/bin/env python3
"""Resident path, option resident_early off / on: microseconds per CG and BiCGStab iteration over box sizes (tolerances off,
best of 3 solves of `iters` iterations).    python tools/resident_ab.py [sizes: 16,32,64,100,128] [iters]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,32,64,100,128").split(",")]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = api.Context(0)
for e in sizes:
    g = mesh.structured_box(e, e, e)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    b = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    line = {"box": e}
    for name, cls in (("cg", api.CgSolver), ("bicgstab", api.BiCgStabSolver)):
        for early in (0, 1):
            ctx.set_option("resident_early", early)
            best, before = None, ctx.counter("resident_solves")
            for _ in range(3):
                s = cls()
                s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
                x = api.DeviceVector(ctx, g.n_cells)
                ctx.sync()
                t = time.perf_counter()
                s.solve(x, b, op)
                ctx.sync()
                dt = (time.perf_counter() - t) / iters * 1e6
                best = dt if best is None else min(best, dt)
            line[f"{name}_{'early' if early else 'late'}_us"] = round(best, 2) if ctx.counter("resident_solves") - before == 3 else None
    ctx.set_option("resident_early", 1)
    print(json.dumps(line), flush=True)
    mat.close()
ctx.close()
