#!/usr/bin/env python3
"""Cost of the multi-rank code path with the network taken out: one rank, a communicator of size 1 (RCCL, then the
peer-window transport -- "ipc": the interior launch sends, the boundary launch reads the window, the reductions'
finishing block all-reduces), a z-periodic
256^3 box whose two halo planes are exchanged with the rank itself, against the plain single-GPU path on the same
box.  What remains at N > 1 beyond this is the latency of the real exchanges."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stormruler_amd import api, mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
only = sys.argv[2] if len(sys.argv) > 2 else ""  # "ipc" / "rccl": that transport alone, few iterations (for a kernel trace)
iters = int(os.environ.get("COMM_ITERS", "60" if only else "400"))
solver = os.environ.get("COMM_SOLVER", "cg")  # "bicgstab": BASELINE config 2's loop


def rate(ctx, mat, g):
    b = api.DeviceVector(ctx, g.n_cells, g.n_halo)
    api.fill_with(b, 1.0)
    best = 0.0
    for _ in range(3):
        x = api.DeviceVector(ctx, g.n_cells, g.n_halo)
        s = api.BiCgStabSolver() if solver == "bicgstab" else api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
        ctx.sync()
        t0 = time.perf_counter()
        s.solve(x, b, api.HipStencilOperator(mat, -1.0, 0.05))
        ctx.sync()
        best = max(best, iters / (time.perf_counter() - t0))
    return best


out = {"n": n, "solver": solver}
ctx = api.Context(0)
g0 = mesh.structured_box(n)
for fmt in (() if only else (4, 3)):
    ctx.set_option("spmv_dict", fmt)
    m0 = api.StencilMatrix.from_face_graph(ctx, g0)
    out[f"plain_fmt{fmt}_it_per_s"] = rate(ctx, m0, g0)
    m0.close()
ctx.close()
loc, send_idx = mesh.periodic_z_local_graph(n, n, n)
# the halo operator: MIXED records by default (format 4 where rows read no halo column, format 3 in the outer
# planes) -- compared with the plain format-4 operator; spmv_mixed = 0 (format 3 throughout) with plain format 3
for transport in ((only,) if only else ("rccl", "ipc")):
    for mixed in ((1,) if only else (1, 0)):
        ctx = api.Context(0)
        ctx.set_option("spmv_mixed", mixed)
        for kv in os.environ.get("COMM_OPTS", "").split(","):  # e.g. COMM_OPTS=rccl_flag_wait=0
            if "=" in kv:
                ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        if transport == "rccl":
            ctx.comm_init(api.Context.comm_unique_id(), 1, 0)
        else:  # peer windows: the single rank maps its own window
            # (both planes go to the one "neighbour" here: twice the default segment)
            ctx.comm_init_ipc(ctx.comm_ipc_export(1, 0, 16 << 20))
        m1 = api.StencilMatrix.from_face_graph(ctx, loc)
        m1.set_halo([0], [0, loc.n_halo], send_idx, [0, loc.n_halo])
        st = m1.stats()
        key = f"{transport}_{'mixed' if mixed else 'fmt3'}"
        out[f"{key}_it_per_s"] = rate(ctx, m1, loc)
        out["interior_groups"], out["groups"], out[f"{key}_paired_rows"] = st["n_interior_slices"], st["n_slices"], st["paired_rows"]
        out[f"{key}_tiled_planes"] = st["tiled_planes"]
        if transport == "rccl":  # where the RCCL path's time goes, by device timestamps (an instrumented solve)
            ctx.set_option("profile_comm", 1)
            rate_iters = iters
            b_ = api.DeviceVector(ctx, loc.n_cells, loc.n_halo)
            api.fill_with(b_, 1.0)
            s_ = api.BiCgStabSolver() if solver == "bicgstab" else api.CgSolver()
            s_.num_iterations, s_.absolute_error_tolerance, s_.relative_error_tolerance = rate_iters, 0.0, 0.0
            s_.solve(api.DeviceVector(ctx, loc.n_cells, loc.n_halo), b_, api.HipStencilOperator(m1, -1.0, 0.05))
            ctx.sync()
            out[f"{key}_comm_breakdown"] = ctx.rccl_profile(rate_iters)
            ctx.set_option("profile_comm", 0)
        if not only:
            base = out["plain_fmt4_it_per_s" if mixed else "plain_fmt3_it_per_s"]
            out[f"{key}_overhead_us_per_iteration"] = (1.0 / out[f"{key}_it_per_s"] - 1.0 / base) * 1e6
        m1.close()
        ctx.close()
print(json.dumps(out))
