#!/usr/bin/env python3
"""Full-size, full-solve parity at the BASELINE sizes: the GPU solve and the CPU oracle's to the default
tolerances (minutes of CPU time; not part of the test suite).  Prints one JSON line per case for profiles/.

    python tools/full_size_parity.py [cg256] [bicgstab256] [gmres128cd]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from stormruler_amd import api, mesh  # noqa: E402

NU, VEL = 1e-2, (1.0, 0.5, 0.25)


def run(case, ctx):
    if case in ("cg256", "bicgstab256"):
        n, kind = 256, case[:-3]
        g = mesh.structured_box(n)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        op = api.HipStencilOperator(mat, -1.0, 0.0)
        ref_op = oracle.StencilOperator(g, -1.0, 0.0)
        cls, kw, name = (api.CgSolver if kind == "cg" else api.BiCgStabSolver), {}, f"{kind.upper()}, 256^3 Poisson"
    elif case == "gmres128cd":
        n, kind = 128, "gmres"
        g = mesh.structured_box(n)
        wi, wo, de = mesh.convection_diffusion_weights(g, NU, VEL)
        mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, g.n_halo, g.inner, g.outer, wi, wo, de)
        op = api.HipStencilOperator(mat, 1.0, 0.0)
        ref_op = oracle.StencilOperator(g, -NU, 0.0, conv=1.0, vel=VEL)
        cls, kw, name = api.GmresSolver, {"num_inner_iterations": 30}, "GMRES(30), 128^3 convection-diffusion"
    else:
        raise SystemExit(f"unknown case {case}")
    b, x = api.DeviceVector(ctx, g.n_cells), api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    s = cls()
    for k, v in kw.items():
        setattr(s, k, v)
    s.record_history = True
    ctx.sync()
    t = time.perf_counter()
    ok = s.solve(x, b, op)
    ctx.sync()
    tg = time.perf_counter() - t
    t = time.perf_counter()
    r = oracle.solve(kind, ref_op, np.ones(g.n_cells), num_inner_iterations=kw.get("num_inner_iterations", 50))
    tc = time.perf_counter() - t
    xg = x.to_numpy()
    m = min(len(s.history), len(r.history), 100)
    st = mat.stats()
    out = {"case": f"{name}, default tolerances", "record_format": "paired rows" if st["paired_rows"] else
           ("dictionary" if st["value_dictionary_size"] else "fp64"), "gpu_converged": bool(ok),
           "cpu_converged": r.converged, "gpu_iterations": int(s.iteration), "cpu_iterations": int(r.iterations),
           "solution_rel_l2_diff": float(np.linalg.norm(xg - r.x) / np.linalg.norm(r.x)),
           "max_rel_history_diff_first_100": float(np.max(np.abs(s.history[:m] - r.history[:m]) / r.history[:m])),
           "gpu_final_rel_residual": s.relative_error, "cpu_final_rel_residual": r.relative_error,
           "gpu_seconds": tg, "cpu_seconds_1_thread": tc, "speedup": tc / tg}
    mat.close()
    return out


if __name__ == "__main__":
    cases = sys.argv[1:] or ["cg256"]
    ctx = api.Context(0)
    for c in cases:
        print(json.dumps(run(c, ctx)), flush=True)
