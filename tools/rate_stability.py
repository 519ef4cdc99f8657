#!/usr/bin/env python3
"""Is the 256^3 CG rate a property of the box, of the process, or of where a solve's vectors land?  Twelve solves in one
process: x allocated anew per solve (the bench's way), then one x reused; the device address of x beside each rate.
    python tools/rate_stability.py [n] [iters]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes  # noqa: E402

from stormruler_amd import api, mesh  # noqa: E402
from stormruler_amd._lib import lib  # noqa: E402


def addr(v):
    p = ctypes.c_void_p()
    lib.storm_hip_vec_device_ptr(v._h, ctypes.byref(p))
    return p.value

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = mesh.structured_box(n)
ctx = api.Context(0)
mat = api.StencilMatrix.from_face_graph(ctx, g)
op = api.HipStencilOperator(mat, -1.0, 0.0)
b = api.DeviceVector(ctx, g.n_cells)
api.fill_with(b, 1.0)


def one(x):
    s = api.CgSolver()
    s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
    api.fill_with(x, 0.0)
    ctx.sync()
    t0 = time.perf_counter()
    s.solve(x, b, op)
    ctx.sync()
    return iters / (time.perf_counter() - t0)


out = {"fresh_x": [], "same_x": [], "iters": iters}
keep = []
for k in range(12):
    x = api.DeviceVector(ctx, g.n_cells)
    out["fresh_x"].append([round(one(x), 1), hex(addr(x))])
    if k % 3 == 0:
        keep.append(x)  # (every third stays allocated: the next one lands elsewhere)
x = api.DeviceVector(ctx, g.n_cells)
out["same_x_address"] = hex(addr(x))
out["b_address"] = hex(addr(b))
for k in range(12):
    out["same_x"].append(round(one(x), 1))
for odd in (iters + 1, iters):
    s = api.CgSolver()
    out[f"same_x_{odd}_iterations"] = []
    for k in range(6):
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = odd, 0.0, 0.0
        api.fill_with(x, 0.0)
        ctx.sync()
        t0 = time.perf_counter()
        s.solve(x, b, op)
        ctx.sync()
        out[f"same_x_{odd}_iterations"].append(round(odd / (time.perf_counter() - t0), 1))
print(json.dumps(out))
