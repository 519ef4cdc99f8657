#!/usr/bin/env python3
"""Does the latency path (rows published by plain write-through stores, gathered behind an all-reduce) stay bitwise
reproducible while ANOTHER stream saturates HBM and the fabric?  (csrc/ticket_device.hpp found an acknowledged store
not yet visible to another XCD under such load.)  Reference: CG / BiCGStab on 64^3 and 80^3 on an idle device; then
the same solves while torch copies 1 GiB buffers back to back on a second stream.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

ctx = api.Context(0)
ctx.set_option("latency_rows", 1 << 21)
out = {}
a = torch.empty(1 << 27, dtype=torch.float64, device="cuda")  # 1 GiB
b = torch.empty_like(a)
a.fill_(1.0)
side = torch.cuda.Stream()
for edge in (64, 80):
    g = mesh.structured_box(edge)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    op = api.HipStencilOperator(mat, -1.0, 0.0)
    bh = api.DeviceVector.from_numpy(ctx, 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells)))
    for name, cls, iters in (("cg", api.CgSolver, 1000), ("bicgstab", api.BiCgStabSolver, 100)):
        def run():
            s = cls()
            s.record_history, s.num_iterations = True, iters
            s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
            x = api.DeviceVector(ctx, g.n_cells)
            s.solve(x, bh, op)
            return np.array(s.history), x.to_numpy()
        ref_h, ref_x = run()
        mismatches, solves, t0 = 0, 0, time.time()
        while time.time() - t0 < 6.0:
            with torch.cuda.stream(side):
                for _ in range(40):
                    b.copy_(a, non_blocking=True)
            for _ in range(4):
                h, x = run()
                solves += 1
                if not (np.array_equal(h, ref_h) and np.array_equal(x, ref_x)):
                    mismatches += 1
            side.synchronize()
        out[f"{name}_{edge}"] = {"solves_under_load": solves, "mismatches": mismatches, "iterations_each": iters}
    mat.close()
print(json.dumps(out))
