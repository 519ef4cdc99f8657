"""Worker of tests/test_gpu_two_ranks.py::test_unstructured_partition_reproduces_the_recorded_reference_run:
the reference's own Triangle mesh (tests/golden/mesh/square_nb.1.*), cut into W parts by recursive coordinate
bisection (general partitioner + general halo plans: arbitrary neighbour sets, not slabs), every part on device 0
over the host-staged transport.  The recorded reference run -- CG on y = x - 1e-2 div grad x, b = sin(3x)cos(7y):
104 iterations, |x|_2 = 27.409299681049 -- must come out of the partitioned device solve."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as td  # noqa: E402

from stormruler_amd import api, dist, io_tetgen, mesh, partition  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))["baseline_md_probe"]["unstructured"]
    g = io_tetgen.read_triangle(os.path.join(ROOT, golden["mesh"]))
    g = mesh.FaceGraph(g.n_cells, 2, g.inner, g.outer, g.area, g.center, g.volume, b_center=np.zeros((0, 2)))  # Neumann
    part = partition.rcb_partition(g.center, world)
    loc = partition.partition_graph(g, part, rank)
    plan = partition.halo_plan(loc, rank)
    ctx = api.Context(0)
    dist.connect_host_staged(ctx)
    mat = api.StencilMatrix.from_face_graph(ctx, loc)
    mat.set_halo(plan.nbr_rank, plan.send_ptr, plan.send_idx, plan.recv_ptr)
    n = loc.n_cells
    c = loc.center[:n]
    b = api.DeviceVector.from_numpy(ctx, np.sin(3 * c[:, 0]) * np.cos(7 * c[:, 1]), n_halo=loc.n_halo)
    x = api.DeviceVector(ctx, n, loc.n_halo)
    s = api.CgSolver()
    assert s.solve(x, b, api.HipStencilOperator(mat, golden["alpha"], golden["beta"]))
    nrm = api.norm_2(x)  # global
    assert abs(s.iteration - golden["iterations"]) <= 1, (s.iteration, golden["iterations"])
    assert abs(nrm - golden["x_norm2"]) <= 1e-8 * golden["x_norm2"], (nrm, golden["x_norm2"])
    owner0 = int(part[0])
    if rank == owner0:  # x[0] of the recorded run lives on the rank that owns global cell 0
        local0 = int(np.nonzero(loc.global_id[:n] == 0)[0][0])
        assert abs(x.to_numpy()[local0] - golden["x0"]) <= 1e-7 * abs(golden["x0"])
    with open(os.path.join(os.environ["STORM_REPORT_DIR"], f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "n_local": int(n), "n_halo": int(loc.n_halo), "nbrs": [int(r) for r in plan.nbr_rank],
                   "iterations": int(s.iteration), "x_norm2": float(nrm)}, f)
    mat.close()
    ctx.close()
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
