#!/usr/bin/env python3
"""256^3 CG rate against WHERE x lies: a dummy allocation of j x `step` MiB in front of x shifts its address; the solver's
work vectors and b stay where they are.  One JSON line per shift.   python tools/placement_probe.py [shifts] [step_MiB] [option=value ...]"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402
from stormruler_amd._lib import lib  # noqa: E402


def addr(v):
    p = ctypes.c_void_p()
    lib.storm_hip_vec_device_ptr(v._h, ctypes.byref(p))
    return p.value


args = [a for a in sys.argv[1:] if "=" not in a]
shifts = int(args[0]) if len(args) > 0 else 70
step = float(args[1]) if len(args) > 1 else 2.0
iters = 150
g = mesh.structured_box(256)
ctx = api.Context(0)
for kv in [a for a in sys.argv[1:] if "=" in a]:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
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


x = api.DeviceVector(ctx, g.n_cells)
one(x), one(x)  # (the work vectors now sit in the context's pool; every later solve reuses them)
del x
ctx.set_option("pool_bytes", 600 << 20)  # room for the four work vectors only: x and the dummies are really freed
for j in range(shifts):
    dummy = api.DeviceVector(ctx, max(1, int(j * step * 131072))) if j else None
    x = api.DeviceVector(ctx, g.n_cells)
    r = [one(x), one(x)]
    print(json.dumps({"shift_MiB": j * step, "x": hex(addr(x)), "b": hex(addr(b)), "it_per_s": [round(v, 1) for v in r]}), flush=True)
    del x, dummy
