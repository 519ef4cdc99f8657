#!/usr/bin/env python3
"""CG and BiCGStab microseconds per iteration over problem sizes: the resident path (csrc/resident.hip: lattice operators, a
box of the lattice per block), the latency path (csrc/latency.hip: any small operator, a kernel per solve) and the
throughput path (csrc/solvers.hip: a kernel per statement) -- which one pays where?  One JSON line per size and solver;
`default_path_us_per_iteration` is what a caller gets with no option set."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, io_tetgen, mesh  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rate(ctx, op, b, n, latency, iters=400, cls=None, mode=None):
    # mode: None = `latency` decides between the latency path alone (2) and the kernel-per-statement path (0);
    #       "default" = the library's own choice; "resident" = the same, counted only if the resident path took it
    ctx.set_option("latency_path", 1 if mode else 2 if latency else 0)
    before = ctx.counter("resident_solves")
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
    if mode == "resident" and ctx.counter("resident_solves") == before:
        return None
    return best


def main():
    ctx = api.Context(0)
    ctx.set_option("latency_rows", 1 << 22)
    cases = [("box", (e, e, e)) for e in (16, 32, 48, 64, 80, 100, 128, 144, 160)] + [("step1", None)]
    for kind, shape in cases:
        if kind == "box":
            g = mesh.structured_box(*shape)
            alpha, beta = -1.0, 0.0
        else:
            g = io_tetgen.read_triangle(os.path.join(ROOT, "tests", "golden", "mesh", "step.1."))
            g = mesh.FaceGraph(g.n_cells, 2, g.inner, g.outer, g.area, g.center, g.volume, b_center=np.zeros((0, 2)))
            alpha, beta = -1e-2, 1.0
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        op = api.HipStencilOperator(mat, alpha, beta)
        b = api.DeviceVector(ctx, g.n_cells)
        api.fill_with(b, 1.0)
        for name, cls in (("cg", api.CgSolver), ("bicgstab", api.BiCgStabSolver)):
            big = g.n_cells > (1 << 21)  # (the latency path's register variants end at 2^21 rows)
            lat = None if big else rate(ctx, op, b, g.n_cells, True, cls=cls)
            thr = rate(ctx, op, b, g.n_cells, False, cls=cls)
            res = rate(ctx, op, b, g.n_cells, True, cls=cls, mode="resident")
            dflt = rate(ctx, op, b, g.n_cells, True, cls=cls, mode="default")
            ctx.set_option("test_disable", 16)  # rows published with write-through stores (round 2) instead of awaited exchanges
            lat_store = None if big else rate(ctx, op, b, g.n_cells, True, cls=cls)
            ctx.set_option("test_disable", 0)
            line = {"solver": name, "mesh": kind if shape is None else "x".join(map(str, shape)), "rows": g.n_cells,
                    "resident_path_us_per_iteration": res, "default_path_us_per_iteration": dflt,
                    "latency_path_us_per_iteration": lat, "latency_path_store_published_us": lat_store,
                    "throughput_path_us_per_iteration": thr, "ratio_throughput_over_default": thr / dflt}
            if name == "bicgstab" and not big:
                ctx.set_option("latency_cache", 0)
                line["latency_path_records_not_cached_us"] = rate(ctx, op, b, g.n_cells, True, cls=cls)
                ctx.set_option("latency_cache", 1)
            print(json.dumps(line), flush=True)
        mat.close()


if __name__ == "__main__":
    main()
