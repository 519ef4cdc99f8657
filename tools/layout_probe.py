#!/usr/bin/env python3
"""Is the fused CG step sensitive to WHERE its vectors lie?  Per trial: a fresh context, random-sized dummy allocations
between the vectors the solve will use (its work vectors are taken from the context's pool: created here, addresses
noted, released into the pool), one 300-iteration CG solve of the 256^3 problem timed.  One JSON line per trial."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402
from stormruler_amd._lib import check, lib  # noqa: E402

n = 256
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
g = mesh.structured_box(n)
N = g.n_cells


def ptr(v):
    p = C.c_void_p()
    check(lib.storm_hip_vec_device_ptr(v._h, C.byref(p)))
    return p.value


for t in range(trials):
    ctx = api.Context(0)
    pads = []

    def pad():
        k = int(rng.integers(0, 64))  # 0 .. 63 units of 256 KiB
        if t > 0 and k:
            pads.append(api.DeviceVector(ctx, k * 32768))

    pad()
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    pad()
    b = api.DeviceVector(ctx, N)
    api.fill_with(b, 1.0)
    pad()
    x = api.DeviceVector(ctx, N)
    work = []
    for _ in range(4):
        pad()
        work.append(api.DeviceVector(ctx, N))
    addr = {"b": ptr(b), "x": ptr(x), "work": [ptr(w) for w in work]}
    del work  # -> the pool; the solve takes them back (last released first)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    best = None
    for _ in range(3):
        api.fill_with(x, 0.0)
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = 300, 0.0, 0.0
        ctx.sync()
        t0 = time.perf_counter()
        s.solve(x, b, op)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 300 * 1e6
        best = dt if best is None else min(best, dt)
    base = min([addr["b"], addr["x"]] + addr["work"])
    print(json.dumps({"trial": t, "us_per_iteration": round(best, 1),
                      "offsets_MiB": {"b": round((addr["b"] - base) / 2**20, 3), "x": round((addr["x"] - base) / 2**20, 3),
                                      "work": [round((a - base) / 2**20, 3) for a in addr["work"]]},
                      "low_bits_KiB": {"x": (addr["x"] % 2**21) // 1024, "work": [(a % 2**21) // 1024 for a in addr["work"]]}}), flush=True)
    del x, b, op, pads
    mat.close()
    ctx.close()
