#!/usr/bin/env python3
"""A/B of the two record formats of the SpMV (fp64 weights vs value-dictionary indices) on one mesh:
bitwise-equal results, interleaved timing rounds, and the CG rate on top of each."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--cg-iters", type=int, default=400)
    args = ap.parse_args()
    g = mesh.structured_box(args.n)
    ctx = api.Context(0)
    mats = {}
    # (format, slices per wave): 0 fp64 records, 1 value dictionary, 2 value + offset dictionaries
    for d in ((0, 1), (1, 2), (2, 1), (2, 2), (2, 4)):
        ctx.set_option("spmv_dict", d[0])
        ctx.set_option("spmv_spw", d[1])
        mats[d] = api.StencilMatrix.from_face_graph(ctx, g)
    ctx.set_option("spmv_dict", 2)
    ctx.set_option("spmv_spw", 0)
    N = g.n_cells
    st = {d: mats[d].stats() for d in mats}
    x = api.DeviceVector.from_numpy(ctx, np.sin(0.37 * np.arange(N)))
    y = {d: api.DeviceVector(ctx, N) for d in mats}
    for d in mats:
        mats[d].apply(-1.0, 0.0, x, y[d])
    same = all(bool(np.array_equal(y[(0, 1)].to_numpy(), y[d].to_numpy())) for d in mats)
    ms = {d: [] for d in mats}
    for _ in range(args.rounds):
        for d in mats:
            for _ in range(5):
                mats[d].apply(-1.0, 0.0, x, y[d])
            ctx.timer_start()
            for _ in range(args.reps):
                mats[d].apply(-1.0, 0.0, x, y[d])
            ms[d].append(ctx.timer_stop() / args.reps)
    b = api.DeviceVector(ctx, N)
    api.fill_with(b, 1.0)
    cg = {}
    import time
    for rnd in range(2):
        for d in mats:
            xs = api.DeviceVector(ctx, N)
            s = api.CgSolver()
            s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = args.cg_iters, 0.0, 0.0
            ctx.sync()
            t0 = time.perf_counter()
            s.solve(xs, b, api.HipStencilOperator(mats[d], -1.0, 0.0))
            ctx.sync()
            cg[d] = {"it_per_s": args.cg_iters / (time.perf_counter() - t0), "final_residual": s.absolute_error}
    alg = 24 * N + 12 * st[(0, 1)]["nnz_offdiag"]
    out = {"n": args.n, "bitwise_equal": same, "algorithmic_bytes": alg}
    for d in mats:
        t = float(np.median(ms[d]))
        fmt = st[d]["record_bytes"] + 16 * N
        out[f"fmt{d[0]}_spw{d[1]}"] = {"ms": t, "rounds": ms[d], "record_bytes": st[d]["record_bytes"],
                                        "dictionary": st[d]["value_dictionary_size"], "offsets": st[d]["offset_dictionary_size"], "format_bytes": fmt,
                                        "format_GBs": fmt / t / 1e6, "algorithmic_GBs": alg / t / 1e6, "cg": cg[d]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
