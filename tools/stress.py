"""Soak test (not in the suite): thousands of back-to-back solves must all converge with ONE iteration count per
(solver, path) -- the reductions are run-to-run reproducible -- and leave no memory behind:
  * 64^3: CG and BiCGStab on the latency path (one cooperative persistent kernel per solve: 1 500 launches each of a grid that
    synchronises through memory -- a lost wake-up would hang here), CG / BiCGStab / GMRES on the throughput path,
    CG with a lambda operator and IDR(4) on the engine;
  * the reference's Triangle mesh step.1 on the latency path;
  * one 60 000-iteration CG run on 128^3 and one 200 000-iteration run on 32^3 (latency path) with the tolerances off."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from stormruler_amd import api, io_tetgen, mesh  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ctx = api.Context(0)
out = {}
g = mesh.structured_box(64)
mat = api.StencilMatrix.from_face_graph(ctx, g)
op = api.HipStencilOperator(mat, -1.0, 0.0)
lam = api.make_operator(lambda y, x: mat.apply(-1.0, 0.0, x, y))
b = api.DeviceVector.from_numpy(ctx, np.ones(g.n_cells))
its = set()
t = time.time()
free0 = torch.cuda.mem_get_info()[0]
cases = [("cg/latency", api.CgSolver, op, 1), ("cg/throughput", api.CgSolver, op, 0),
         ("bicgstab/latency", api.BiCgStabSolver, op, 1), ("bicgstab/throughput", api.BiCgStabSolver, op, 0),
         ("gmres", api.GmresSolver, op, 0), ("cg/engine-lambda", api.CgSolver, lam, 0), ("idrs/engine", api.IdrsSolver, op, 0)]
n_rounds = 1500
for k in range(n_rounds):
    for name, cls, operator, latency in cases:
        if name == "idrs/engine" and k % 10:
            continue
        ctx.set_option("latency_path", 2 if latency else 0)
        api.rng_reset()
        x = api.DeviceVector(ctx, g.n_cells)
        s = cls()
        assert s.solve(x, b, operator), (name, k)
        its.add((name, s.iteration))
ctx.set_option("latency_path", 1)
out["repeat_solves"] = {"rounds": n_rounds, "seconds": time.time() - t, "distinct_iteration_counts": sorted(its),
                        "one_count_per_case": len(its) == len(cases), "free_mem_delta": free0 - torch.cuda.mem_get_info()[0]}
tri = io_tetgen.read_triangle(os.path.join(ROOT, "tests", "golden", "mesh", "step.1."))
tri = mesh.FaceGraph(tri.n_cells, 2, tri.inner, tri.outer, tri.area, tri.center, tri.volume, b_center=np.zeros((0, 2)))
mt = api.StencilMatrix.from_face_graph(ctx, tri)
bt = api.DeviceVector.from_numpy(ctx, np.sin(3 * tri.center[:, 0]) * np.cos(7 * tri.center[:, 1]))
its_t = set()
t = time.time()
for k in range(300):
    x = api.DeviceVector(ctx, tri.n_cells)
    s = api.CgSolver()
    assert s.solve(x, bt, api.HipStencilOperator(mt, -1e-2, 1.0))
    its_t.add(s.iteration)
out["step1_latency_path"] = {"solves": 300, "seconds": time.time() - t, "iteration_counts": sorted(its_t)}
for edge, iters, key in ((128, 60000, "long_run_128_throughput"), (32, 200000, "long_run_32_latency")):
    g2 = mesh.structured_box(edge)
    mat2 = api.StencilMatrix.from_face_graph(ctx, g2)
    b2 = api.DeviceVector.from_numpy(ctx, np.ones(g2.n_cells))
    x2 = api.DeviceVector(ctx, g2.n_cells)
    s = api.CgSolver()
    s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = iters, 0.0, 0.0
    t = time.time()
    s.solve(x2, b2, api.HipStencilOperator(mat2, -1.0, 0.0))
    ctx.sync()
    out[key] = {"iterations": s.iteration, "seconds": time.time() - t, "final_abs_err": s.absolute_error,
                "finite": bool(np.isfinite(x2.to_numpy()).all())}
    mat2.close()
# round 3: the lattice kernels at full size -- 40 full CG solves of the 256^3 problem (marching fused step, ticketed
# reductions): one iteration count, bitwise the same solution every time
g3 = mesh.structured_box(256)
mat3 = api.StencilMatrix.from_face_graph(ctx, g3)
b3 = api.DeviceVector.from_numpy(ctx, np.ones(g3.n_cells))
its3, sums = set(), set()
t = time.time()
for k in range(40):
    x3 = api.DeviceVector(ctx, g3.n_cells)
    s = api.CgSolver()
    assert s.solve(x3, b3, api.HipStencilOperator(mat3, -1.0, 0.0))
    its3.add(s.iteration)
    sums.add(float(api.norm_2(x3)))
out["cg_256_marching_step"] = {"solves": 40, "seconds": time.time() - t, "iteration_counts": sorted(its3),
                               "distinct_solution_norms": len(sums), "tiled_planes": mat3.stats()["tiled_planes"]}
mat3.close()
print(json.dumps(out))
