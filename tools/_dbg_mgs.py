import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh
g = mesh.structured_box(128)
wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
ctx = api.Context(0)
ctx.set_option("resident_profile", 1)
mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
b = api.DeviceVector(ctx, g.n_cells); api.fill_with(b, 1.0)
s = api.GmresSolver(); s.num_inner_iterations, s.num_iterations = 30, 60
s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
x = api.DeviceVector(ctx, g.n_cells)
s.solve(x, b, api.HipStencilOperator(mat, 1.0, 0.0))
names = ["update_w", "rows_and_dots", "allreduce", "tail", "x"]
print(json.dumps({k: {n: round(ctx.counter(f"resident_phase_{k}_{i}") * 0.01, 2) for i, n in enumerate(names)} for k in ("mean", "max")}))
import numpy as np
rows = np.array([ctx.counter(f"resident_phase_block_{b}_1") for b in range(256)]) * 0.01
ar = np.array([ctx.counter(f"resident_phase_block_{b}_2") for b in range(256)]) * 0.01
print("rows+dots per block (us, k=29 chain): min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (rows.min(), np.percentile(rows, 10), np.median(rows), np.percentile(rows, 90), rows.max()))
print("allreduce per block: min %.1f median %.1f max %.1f" % (ar.min(), np.median(ar), ar.max()))
order = np.argsort(rows)
print("slowest blocks:", order[-12:].tolist(), "fastest:", order[:12].tolist())
print("by b%8 mean rows:", [round(float(rows[np.arange(256) % 8 == x].mean()), 1) for x in range(8)])
