#!/usr/bin/env python3
"""Kernel launches per iteration of every engine solver (lambda operator, step.1): run under
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/census -- python3 tools/engine_launch_census.py <solver>
and divide the calls of tools/kernel_stats.py by the 400 iterations."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stormruler_amd import api, io_tetgen, mesh  # noqa: E402
from tools.solver_paths import SOLVERS, make  # noqa: E402

kind = sys.argv[1]
ctx = api.Context(0)
g = io_tetgen.read_triangle(os.path.join(ROOT, "tests", "golden", "mesh", "step.1."))
g = mesh.FaceGraph(g.n_cells, 2, g.inner, g.outer, g.area, g.center, g.volume, b_center=np.zeros((0, 2)))
mat = api.StencilMatrix.from_face_graph(ctx, g)
lam = api.make_operator(lambda y, x: mat.apply(-1e-2, 1.0, x, y))
b = api.DeviceVector.from_numpy(ctx, np.sin(3 * g.center[:, 0]) * np.cos(7 * g.center[:, 1]))
s = make(kind)
s.num_iterations = 400
x = api.DeviceVector(ctx, g.n_cells)
s.solve(x, b, lam)
ctx.sync()
print(kind, s.iteration)
