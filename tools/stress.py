"""Soak test (not in the suite): 4 500 back-to-back solves (CG / BiCGStab / GMRES on 64^3) -- every solve must
converge with the same iteration count -- and one 60 000-iteration CG run on 128^3 with the tolerances off."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from stormruler_amd import api, mesh  # noqa: E402
ctx = api.Context(0)
out = {}
g = mesh.structured_box(64)
mat = api.StencilMatrix.from_face_graph(ctx, g)
op = api.HipStencilOperator(mat, -1.0, 0.0)
b = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells))
its = set(); t = time.time()
free0 = torch.cuda.mem_get_info()[0]
for k in range(1500):
    for cls in (api.CgSolver, api.BiCgStabSolver, api.GmresSolver):
        x = api.DeviceVector(ctx, g.n_cells)
        s = cls(); ok = s.solve(x, b, op)
        assert ok
        its.add((cls.__name__, s.iteration))
out["repeat_solves"] = {"count": 4500, "seconds": time.time() - t, "distinct_iteration_counts": sorted(its), "free_mem_delta": free0 - torch.cuda.mem_get_info()[0]}
g = mesh.structured_box(128)
mat2 = api.StencilMatrix.from_face_graph(ctx, g)
b2 = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells)); x2 = api.DeviceVector(ctx, g.n_cells)
s = api.CgSolver(); s.num_iterations = 60000; s.absolute_error_tolerance = 0.0; s.relative_error_tolerance = 0.0
t = time.time(); s.solve(x2, b2, api.HipStencilOperator(mat2, -1.0, 0.0)); ctx.sync()
out["long_run"] = {"iterations": s.iteration, "seconds": time.time() - t, "final_abs_err": s.absolute_error, "finite": bool(np.isfinite(x2.to_numpy()).all())}
print(json.dumps(out))
