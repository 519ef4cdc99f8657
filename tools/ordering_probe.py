"""SpMV rate of a scrambled 256^3 (or --edge) box under different cell orderings: reverse Cuthill-McKee, Morton on the
cell centres, lexicographic on the cell centres.  One JSON line per ordering."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402


def morton_order(center, bits=10):
    lo, hi = center.min(axis=0), center.max(axis=0)
    q = np.minimum(((center - lo) / (hi - lo + 1e-300) * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    key = np.zeros(center.shape[0], np.uint64)
    for b in range(bits):
        for d in range(center.shape[1]):
            key |= ((q[:, d] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * center.shape[1] + d)
    return np.argsort(key, kind="stable").astype(np.int64)


def lex_order(center):
    return np.lexsort((center[:, 0], center[:, 1], center[:, 2])).astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=256)
    ap.add_argument("--jitter", type=int, default=0)
    args = ap.parse_args()
    n = args.edge
    g0 = mesh.structured_box(n)
    if args.jitter:
        g0 = mesh.jitter_geometry(g0, 1.0 / n)
    N = g0.n_cells
    gs = mesh.permute_cells(g0, mesh.random_permutation(N))
    ctx = api.Context(0)
    b = api.DeviceVector(ctx, N)
    api.fill_with(b, 1.0)
    for name, fn in (("rcm", lambda: mesh.rcm_ordering(gs)), ("morton", lambda: morton_order(gs.center[:N])),
                     ("lexicographic", lambda: lex_order(gs.center[:N]))):
        t0 = time.time()
        order = fn()
        t_order = time.time() - t0
        g = mesh.permute_cells(gs, order)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        st = mat.stats()
        op = api.HipStencilOperator(mat, -1.0, 0.0)
        ctx.set_option("profile_spmv", 1)
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = 100, 0.0, 0.0
        s.solve(api.DeviceVector(ctx, N), b, op)
        samples = ctx.spmv_profile_samples()
        ctx.set_option("profile_spmv", 0)
        t1 = time.perf_counter()
        s.solve(api.DeviceVector(ctx, N), b, op)
        ctx.sync()
        dt = time.perf_counter() - t1
        bytes_ = st["record_bytes"] + 16 * N
        ms = float(np.median(samples))
        print(json.dumps({"ordering": name, "order_seconds": round(t_order, 2), "paired_rows": st["paired_rows"],
                          "value_dictionary_size": st["value_dictionary_size"], "tiled_planes": st["tiled_planes"],
                          "record_bytes_per_row": st["record_bytes"] / N, "spmv_ms_median": ms,
                          "frac_of_8TBs_on_streamed_bytes": bytes_ / (ms * 1e-3) / 8e12, "cg_iter_per_s": 100 / dt,
                          "max_column_distance": int(np.abs(g.inner - g.outer).max())}), flush=True)
        mat.close()
        del g
    ctx.close()


if __name__ == "__main__":
    main()
