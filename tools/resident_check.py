"""Resident path (csrc/resident.hip) against the throughput path on lattice boxes: same convergence rule, counters and
histories, same solutions; and the time per iteration of both (fixed iteration counts: tolerances 0).

    python tools/resident_check.py [--shapes 64,100,128] [--iters 300] [--bicgstab] > gpurun_out/resident_check.jsonl
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402


def solve(ctx, cls, op, b_host, resident, **knobs):
    ctx.set_option("resident_path", int(resident))
    ctx.set_option("latency_path", 1 if resident else 0)  # (0: the kernel-per-statement path)
    s = cls()
    s.record_history = True
    for k, v in knobs.items():
        setattr(s, k, v)
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, b_host.size)
    ok = s.solve(x, b, op)
    return ok, s, x.to_numpy()


def timed(ctx, cls, op, b_host, resident, iters, repeats=3):
    ctx.set_option("resident_path", int(resident))
    ctx.set_option("latency_path", 1 if resident else 0)
    b, x = api.DeviceVector.from_numpy(ctx, b_host), api.DeviceVector(ctx, b_host.size)
    best = {}
    for k in (0, iters):
        ts = []
        for _ in range(repeats):
            s = cls()
            s.num_iterations = k
            s.absolute_error_tolerance = 0.0
            s.relative_error_tolerance = 0.0
            api.fill_with(x, 0.0)
            ctx.sync()
            t0 = time.perf_counter()
            s.solve(x, b, op)
            ctx.sync()
            ts.append(time.perf_counter() - t0)
        best[k] = min(ts)
    return 1e6 * (best[iters] - best[0]) / iters, 1e6 * best[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="24,64,100,128")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--bicgstab", action="store_true")
    args = ap.parse_args()
    ctx = api.Context(0)
    cls = api.BiCgStabSolver if args.bicgstab else api.CgSolver
    for tok in args.shapes.split(","):
        dims = [int(v) for v in tok.split("x")]
        g = mesh.structured_box(*dims)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        op = api.HipStencilOperator(mat, -1.0, 0.0)
        b_host = 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells))
        r0 = ctx.counter("resident_solves")
        ok_r, s_r, x_r = solve(ctx, cls, op, b_host, 1)
        taken = ctx.counter("resident_solves") - r0
        ok_t, s_t, x_t = solve(ctx, cls, op, b_host, 0)
        m = min(len(s_r.history), len(s_t.history))
        hist = float(np.max(np.abs(np.asarray(s_r.history[:m]) / np.asarray(s_t.history[:m]) - 1.0)))
        rec = {"shape": dims, "rows": g.n_cells, "resident_taken": int(taken), "ok": [bool(ok_r), bool(ok_t)],
               "iterations": [int(s_r.iteration), int(s_t.iteration)], "history_rel": hist,
               "x_rel": float(np.linalg.norm(x_r - x_t) / np.linalg.norm(x_t)), "fallback": int(s_r.path_fallback)}
        if taken:
            us_r, fixed_r = timed(ctx, cls, op, b_host, 1, args.iters)
            us_t, fixed_t = timed(ctx, cls, op, b_host, 0, args.iters)
            rec.update({"us_per_iteration": {"resident": round(us_r, 2), "throughput": round(us_t, 2)},
                        "fixed_us": {"resident": round(fixed_r, 1), "throughput": round(fixed_t, 1)}})
        print(json.dumps(rec), flush=True)
        mat.close()
    ctx.close()


if __name__ == "__main__":
    main()
