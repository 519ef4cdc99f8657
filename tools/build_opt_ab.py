#!/usr/bin/env python3
"""Interleaved A/B of a BUILD-time library option (one operator per value) on the SpMV time and the CG rate:
    tools/build_opt_ab.py <key> <v1,v2,...> [n] [cg iterations]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

key, values = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 400
g = mesh.structured_box(n)
ctx = api.Context(0)
mats = {}
for v in values:
    ctx.set_option(key, v)
    mats[v] = api.StencilMatrix.from_face_graph(ctx, g)
N = g.n_cells
xs = [api.DeviceVector.from_numpy(ctx, np.sin(0.37 * np.arange(N) + i)) for i in range(3)]
ys = {v: api.DeviceVector(ctx, N) for v in values}
b = api.DeviceVector(ctx, N)
api.fill_with(b, 1.0)
ref = None
res = {v: {"spmv_ms": [], "cg_it_per_s": []} for v in values}
for rnd in range(4):
    for v in values:
        for i in range(3):
            mats[v].apply(-1.0, 0.0, xs[i], ys[v])
        ctx.timer_start()
        for i in range(30):
            mats[v].apply(-1.0, 0.0, xs[i % 3], ys[v])  # rotating inputs: no Infinity-Cache help on x
        res[v]["spmv_ms"].append(round(ctx.timer_stop() / 30, 5))
        mats[v].apply(-1.0, 0.0, xs[0], ys[v])
        yh = ys[v].to_numpy()
        ref = yh if ref is None else ref
        assert np.array_equal(yh, ref), v
        x = api.DeviceVector(ctx, N)
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
        ctx.sync()
        t0 = time.perf_counter()
        s.solve(x, b, api.HipStencilOperator(mats[v], -1.0, 0.0))
        ctx.sync()
        res[v]["cg_it_per_s"].append(round(iters / (time.perf_counter() - t0), 1))
print(json.dumps({str(k): v for k, v in res.items()}))
