#!/usr/bin/env python3
"""CG and BiCGStab microseconds per iteration, latency path (csrc/latency.hip) against throughput path (csrc/solvers.hip), over
problem sizes -- where does the cooperative persistent kernel stop paying?  One JSON line per size."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, io_triangle, mesh  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rate(ctx, op, b, n, latency, iters=400, cls=None):
    ctx.set_option("latency_path", 2 if latency else 0)
    best = None
    for _ in range(3):
        s = (cls or api.CgSolver)()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
        x = api.DeviceVector(ctx, n)
        ctx.sync()
        t = time.perf_counter()
        s.solve(x, b, op)
        ctx.sync()
        dt = (time.perf_counter() - t) / iters * 1e6
        best = dt if best is None else min(best, dt)
    ctx.set_option("latency_path", 1)
    return best


def main():
    ctx = api.Context(0)
    ctx.set_option("latency_rows", 1 << 22)
    cases = [("box", (e, e, e)) for e in (16, 32, 48, 64, 80, 100, 128)] + [("step1", None)]
    for kind, shape in cases:
        if kind == "box":
            g = mesh.structured_box(*shape)
            alpha, beta = -1.0, 0.0
        else:
            g = io_triangle.read_triangle(os.path.join(ROOT, "tests", "golden", "mesh", "step.1."))
            g = mesh.FaceGraph(g.n_cells, 2, g.inner, g.outer, g.area, g.center, g.volume, b_center=np.zeros((0, 2)))
            alpha, beta = -1e-2, 1.0
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        op = api.HipStencilOperator(mat, alpha, beta)
        b = api.DeviceVector(ctx, g.n_cells)
        api.fill_with(b, 1.0)
        for name, cls in (("cg", api.CgSolver), ("bicgstab", api.BiCgStabSolver)):
            lat, thr = rate(ctx, op, b, g.n_cells, True, cls=cls), rate(ctx, op, b, g.n_cells, False, cls=cls)
            ctx.set_option("latency_publish", 0)  # rows published with write-through stores (round 2) instead of awaited exchanges
            lat_store = rate(ctx, op, b, g.n_cells, True, cls=cls)
            ctx.set_option("latency_publish", 1)
            line = {"solver": name, "mesh": kind if shape is None else "x".join(map(str, shape)), "rows": g.n_cells,
                    "latency_path_us_per_iteration": lat, "latency_path_store_published_us": lat_store,
                    "throughput_path_us_per_iteration": thr, "ratio": thr / lat}
            if name == "bicgstab":
                ctx.set_option("latency_cache", 0)
                line["latency_path_records_not_cached_us"] = rate(ctx, op, b, g.n_cells, True, cls=cls)
                ctx.set_option("latency_cache", 1)
            print(json.dumps(line), flush=True)
        mat.close()


if __name__ == "__main__":
    main()
