#!/usr/bin/env python3
"""BASELINE config 4's shape on one GPU: GMRES(30) on the 128^3 convection-diffusion operator, iterations per second
(fixed 600 iterations, tolerances off), per option set:  python tools/convdiff_rate.py "test_disable=32" "test_disable=0" """
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from stormruler_amd import api, mesh  # noqa: E402
from test_gpu_convdiff import NU, VEL  # noqa: E402

g = mesh.structured_box(128)
wi, wo, de = mesh.convection_diffusion_weights(g, NU, VEL)
for spec in sys.argv[1:] or [""]:
    ctx = api.Context(0)
    for kv in [s for s in spec.split(",") if s]:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
    b = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    best = None
    for _ in range(4):
        s = api.GmresSolver()
        s.num_inner_iterations, s.num_iterations = 30, 600
        s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
        x = api.DeviceVector(ctx, g.n_cells)
        ctx.sync()
        t = time.perf_counter()
        s.solve(x, b, api.HipStencilOperator(mat, 1.0, 0.0))
        ctx.sync()
        dt = (time.perf_counter() - t) / 600
        best = dt if best is None else min(best, dt)
    print(json.dumps({"options": spec, "us_per_iteration": round(best * 1e6, 1), "iterations_per_s": round(1.0 / best)}), flush=True)
    mat.close()
    ctx.close()
