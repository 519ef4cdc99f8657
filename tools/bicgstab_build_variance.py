"""BiCGStab, 256^3 Poisson, full solve on the two CPU builds of the oracle (strict / FMA contraction): how far
the iteration count and the solution move with rounding alone (minutes of CPU time)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from stormruler_amd import mesh
g = mesh.structured_box(256)
out = {}
xs = {}
for variant in ("strict", "fma"):
    t = time.time()
    r = oracle.solve("bicgstab", oracle.StencilOperator(g, -1.0, 0.0, variant=variant), np.ones(g.n_cells), variant=variant)
    out[variant] = {"iterations": int(r.iterations), "rel": r.relative_error, "seconds": time.time() - t}
    xs[variant] = r.x
out["solution_rel_l2_diff_between_cpu_builds"] = float(np.linalg.norm(xs["strict"] - xs["fma"]) / np.linalg.norm(xs["strict"]))
print(json.dumps(out))
