#!/usr/bin/env python3
"""Where the longest Gram-Schmidt chain of a GMRES(30) cycle (30 vectors, ten groups of three) spends its time at 128^3
(BASELINE config 4; option resident_profile: the chain kernel times its phases with the 100 MHz counter): the update of w,
the group's rows landing + the dot products, the all-reduce (with the next group's request inside it), the tail.
Mean and max over the blocks, microseconds for the whole chain."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from stormruler_amd import api, mesh
from test_gpu_convdiff import NU, VEL
g = mesh.structured_box(128)
wi, wo, de = mesh.convection_diffusion_weights(g, NU, VEL)
ctx = api.Context(0)
ctx.set_option("resident_profile", 1)
mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
b = api.DeviceVector(ctx, g.n_cells); api.fill_with(b, 1.0)
s = api.GmresSolver(); s.num_inner_iterations, s.num_iterations = 30, 120
s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
x = api.DeviceVector(ctx, g.n_cells)
s.solve(x, b, api.HipStencilOperator(mat, 1.0, 0.0)); ctx.sync()
out = {}
for kind in ("mean", "max"):
    out[kind] = [round(ctx.counter(f"resident_phase_{kind}_{k}") * 0.01, 2) for k in range(4)]
print(json.dumps({"phases_us (update of w, rows landed + dot products, all-reduce, tail) of the 30-vector chain (10 groups)": out}))
