#!/usr/bin/env python3
"""A/B of library options on the bench problem (256^3 Poisson, CG / BiCGStab with the tolerances off), interleaved rounds:

    python tools/cg_ab.py "spmv_canon_tile=0" "spmv_canon_tile=4" "spmv_canon_tile=2" [--solver cg] [--iters 400]

Per option set: iterations/s (best and median of the rounds), the SpMV's mean launch time from the library's own
HIP-event pairs (option profile_spmv), and the final residual (must agree to rounding between sets).  Options that
shape the operator's records (spmv_dict, spmv_canon_tile, ...) are applied before the operator is built."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sets", nargs="+", help="comma-separated key=value lists ('' = defaults)")
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--solver", default="cg", choices=["cg", "bicgstab"])
    a = ap.parse_args()
    g = mesh.structured_box(a.n)
    N = g.n_cells
    runs = []
    for spec in a.sets:
        ctx = api.Context(0)
        for kv in [s for s in spec.split(",") if s]:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        b = api.DeviceVector(ctx, N)
        api.fill_with(b, 1.0)
        runs.append({"spec": spec, "ctx": ctx, "mat": mat, "b": b, "rates": [], "stats": mat.stats()})

    def solve(r, iters):
        s = api.CgSolver() if a.solver == "cg" else api.BiCgStabSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
        x = api.DeviceVector(r["ctx"], N)
        r["ctx"].sync()
        t0 = time.perf_counter()
        s.solve(x, r["b"], api.HipStencilOperator(r["mat"], -1.0, 0.0))
        r["ctx"].sync()
        return iters / (time.perf_counter() - t0), s.absolute_error

    for r in runs:
        solve(r, 300)  # spin-up
    for _ in range(a.rounds):
        for r in runs:
            rate, res = solve(r, a.iters)
            r["rates"].append(rate)
            r["residual"] = res
    for r in runs:
        r["ctx"].set_option("profile_spmv", 1)
        solve(r, 50)
        launches, total_ms, min_ms = r["ctx"].spmv_profile()
        r["ctx"].set_option("profile_spmv", 0)
        st = r["stats"]
        fmt_bytes = st["record_bytes"] + 16 * N
        ms = total_ms / launches
        print(json.dumps({"options": r["spec"], "solver": a.solver, "n": a.n, "it_per_s_best": max(r["rates"]),
                          "it_per_s_median": float(np.median(r["rates"])), "us_per_iteration_median": 1e6 / float(np.median(r["rates"])),
                          "spmv_avg_launch_ms": ms, "spmv_min_launch_ms": min_ms, "spmv_launches": launches,
                          "spmv_streamed_GBs": fmt_bytes / ms / 1e6, "spmv_frac_of_8TBs": fmt_bytes / ms / 1e6 / 8000.0,
                          "tiled_planes": st["tiled_planes"], "spmv_blocks": st["spmv_blocks"], "paired_rows": st["paired_rows"],
                          "final_residual": r["residual"]}), flush=True)
    for r in runs:
        r["mat"].close()
        r["ctx"].close()


if __name__ == "__main__":
    main()
