#!/usr/bin/env python3
"""GMRES(30) / CG / BiCGStab microseconds per iteration on small and mid-size boxes against a library option
(default: test_disable = 32 / 0, i.e. cooperative launches against ordinary ones): python tools/gmres_ab.py [sizes:32,64,128] [solvers:gmres30,cg,bicgstab] [key=value ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

argv = sys.argv[1:]
sizes = [int(v) for a in argv if a.startswith("sizes:") for v in a[6:].split(",")] or [32, 64, 128]
wanted = [v for a in argv if a.startswith("solvers:") for v in a[8:].split(",")] or ["gmres30", "cg", "bicgstab"]
sets = [a for a in argv if ":" not in a] or ["test_disable=32", "test_disable=0"]
for n in sizes:
    g = mesh.structured_box(n)
    row = {"n": n}
    for spec in sets:
        ctx = api.Context(0)
        for kv in [s for s in spec.split(",") if s]:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        b = api.DeviceVector(ctx, g.n_cells)
        api.fill_with(b, 1.0)
        for name, cls, iters in (("gmres30", api.GmresSolver, 300), ("cg", api.CgSolver, 600), ("bicgstab", api.BiCgStabSolver, 400)):
            if name not in wanted:
                continue
            if n > 128:
                iters = max(60, iters // 5)
            best = None
            for _ in range(4):
                s = cls()
                if name == "gmres30":
                    s.num_inner_iterations = 30
                s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
                x = api.DeviceVector(ctx, g.n_cells)
                ctx.sync()
                t = time.perf_counter()
                s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.0))
                ctx.sync()
                dt = (time.perf_counter() - t) / iters * 1e6
                best = dt if best is None else min(best, dt)
            row[f"{name} {spec}"] = round(best, 1)
        mat.close()
        ctx.close()
    print(json.dumps(row), flush=True)
