"""Worker of tests/test_gpu_two_ranks.py::test_cavity_on_four_ranks: BASELINE config 5's layout (lid-driven cavity,
pressure-Poisson CG every step, 4 ranks) with all ranks on device 0 over the host-staged transport.  Every rank
steps its z-slab of the projection scheme and compares, step by step, with the SAME scheme run unpartitioned on a
second (communicator-free) context of the same GPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as td  # noqa: E402

from stormruler_amd import api, cavity, dist, partition  # noqa: E402


def main():
    n, steps = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    assert n % world == 0
    ctx = api.Context(0)
    dist.connect_host_staged(ctx)
    loc, plan = partition.slab_partition(n, n, n // world, world, rank)
    part = cavity.CavityProjection(ctx, n, nu=0.01, graph=loc, plan=plan)
    ref_ctx = api.Context(0)  # no communicator: the unpartitioned problem
    ref = cavity.CavityProjection(ref_ctx, n, nu=0.01)
    gid = loc.global_id[: loc.n_cells]
    report = {"rank": rank, "world": world, "steps": []}
    for s in range(steps):
        it_p, _, ok_p = part.step()
        it_r, _, ok_r = ref.step()
        assert ok_p and ok_r
        assert abs(it_p - it_r) <= max(2, int(0.1 * it_r)), (s, it_p, it_r)
        worst = 0.0
        for d in range(3):
            u_p, u_r = part.u[d].to_numpy(), ref.u[d].to_numpy()[gid]
            scale = max(np.abs(ref.u[d].to_numpy()).max(), 1e-300)
            worst = max(worst, np.abs(u_p - u_r).max() / scale)
        # (the Neumann pressure is defined up to a constant: it is compared through the corrected velocity)
        assert worst <= 1e-6, (s, worst)
        report["steps"].append([int(it_p), int(it_r), float(worst)])
    # the (global, all-reduced) divergence norm of the partitioned field equals the unpartitioned run's
    div_p, div_r = part.divergence_norm(), ref.divergence_norm()
    assert abs(div_p - div_r) <= 1e-6 * max(div_r, 1e-12), (div_p, div_r)
    report["divergence"] = [float(div_p), float(div_r)]
    with open(os.path.join(os.environ["STORM_REPORT_DIR"], f"rank{rank}.json"), "w") as f:
        json.dump(report, f)
    ctx.close()
    ref_ctx.close()
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
