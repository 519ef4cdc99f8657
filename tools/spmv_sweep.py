#!/usr/bin/env python3
"""A/B harness for the SpMV kernel: one mesh, interleaved rounds of kernel options (guide rule 24)."""
import argparse
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--configs", default="0:1:0:0,0:1:0:1,1:1:0:0,0:0:0:0")
    ap.add_argument("--ordering", default="natural")
    ap.add_argument("--rotate", type=int, default=1, help="number of distinct x/y vector pairs cycled through")
    args = ap.parse_args()
    g = mesh.structured_box(args.n)
    if args.ordering == "tile":
        g = mesh.permute_cells(g, mesh.tile_ordering(args.n, args.n, args.n, 16, 16))
    elif args.ordering == "random":
        g = mesh.permute_cells(g, mesh.random_permutation(g.n_cells))
    elif args.ordering == "rcm":
        g = mesh.permute_cells(g, mesh.random_permutation(g.n_cells))
        g = mesh.permute_cells(g, mesh.rcm_ordering(g))
    ctx = api.Context(0)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    st = mat.stats()
    N = g.n_cells
    bytes_alg = 24 * N + 12 * st["nnz_offdiag"]
    xs = [api.DeviceVector.from_numpy(ctx, np.sin(0.37 * np.arange(N))) for _ in range(args.rotate)]
    ys = [api.DeviceVector(ctx, N) for _ in range(args.rotate)]
    x, y = xs[0], ys[0]
    cfgs = [tuple(int(t) for t in c.split(":")) for c in args.configs.split(",")]
    res = {c: [] for c in cfgs}
    ref = None
    for _ in range(args.rounds):
        for c in cfgs:
            variant, nt, bpc = c[:3]
            ctx.set_option("spmv_xcd_remap", c[3] if len(c) > 3 else 1)
            ctx.set_option("spmv_variant", variant)
            ctx.set_option("nontemporal", nt)
            for _ in range(3):
                mat.apply(-1.0, 0.0, x, y)
            ctx.timer_start()
            for i in range(args.reps):
                mat.apply(-1.0, 0.0, xs[i % args.rotate], ys[i % args.rotate])
            ms = ctx.timer_stop() / args.reps
            mat.apply(-1.0, 0.0, x, y)
            res[c].append(ms)
            yh = y.to_numpy()
            if ref is None:
                ref = yh
            assert np.abs(yh - ref).max() <= 1e-12 * np.abs(ref).max(), c
    for c in cfgs:
        med = float(np.median(res[c]))
        print(json.dumps({"variant": c[0], "nt": c[1], "blocks_per_cu": c[2], "xcd": c[3] if len(c) > 3 else 1, "ms_median": med, "ms_min": min(res[c]),
                          "GBs": bytes_alg / med / 1e6, "frac_of_8TBs": bytes_alg / med / 1e6 / 8000}))


if __name__ == "__main__":
    main()
