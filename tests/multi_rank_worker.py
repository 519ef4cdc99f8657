"""Worker of tests/test_gpu_two_ranks.py: one rank of a W-rank run whose ranks all share device 0.

Launched by torch.distributed.run with the gloo backend.  Every rank builds its z-slab of an nx*ny*(nzl*W) box
on the GPU, connects the library's HOST-STAGED transport (halo planes and reduction scalars travel through gloo) or,
with STORM_TRANSPORT=ipc, its PEER-WINDOW transport (hipIpc-mapped device memory, direct stores, polled flags),
and checks the partitioned device path -- SpMV with interior/boundary split, fused dots, CG / BiCGStab / GMRES
device loops with their all-reduces -- against the CPU oracle on the UNPARTITIONED global mesh.  Everything the
8-GPU run does except the RCCL calls themselves (those: tests/test_gpu_comm.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as td  # noqa: E402

from oracle import oracle  # noqa: E402
from stormruler_amd import api, dist, mesh, partition  # noqa: E402


def main():
    nx, ny, nzl = (int(v) for v in sys.argv[1:4])
    dist.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    ctx = api.Context(0)
    if os.environ.get("STORM_TRANSPORT", "host") == "ipc":
        dist.connect_ipc(ctx)  # peer windows: direct stores into the receiver's memory, polled flags
    else:
        dist.connect_host_staged(ctx)
    assert (ctx.n_ranks, ctx.rank) == (world, rank)

    loc, plan = partition.slab_partition(nx, ny, nzl, world, rank)
    n = loc.n_cells
    gid = loc.global_id[:n]
    # the same global problem, unpartitioned, on the CPU oracle (every rank computes it: small)
    lengths = (1.0, ny / nx, nzl * world / nx)
    glob = mesh.structured_box(nx, ny, nzl * world, lengths=lengths)
    assert glob.n_cells == n * world
    ref_op = oracle.StencilOperator(glob, -1.0, 0.0)
    x_glob = np.sin(0.37 * np.arange(glob.n_cells))
    y_ref = ref_op.apply(x_glob)
    report = {"rank": rank, "world": world, "n_local": int(n), "n_halo": int(loc.n_halo), "nbrs": [int(r) for r in plan.nbr_rank]}

    # format 4 = the mixed records of a slab (canonical rows inside, format 3 in the planes that read halo columns), with
    # the lattice kernels -- tiles, the marching CG step -- forced onto these small slabs where their geometry allows
    ctx.set_option("spmv_canon_tile_min_rows", 0)
    for fmt in (0, 2, 3, 4):
        ctx.set_option("spmv_dict", fmt)
        mat = api.StencilMatrix.from_face_graph(ctx, loc)
        ctx.set_option("spmv_dict", 4)
        st = mat.stats()
        assert (st["offset_dictionary_size"] > 0) == (fmt >= 2), st
        assert (fmt >= 3 or not st["paired_rows"]) and (st["paired_rows"] or fmt < 3 or nx % 2 == 1), st
        if fmt == 4 and nx % 2 == 0:
            # (mixed records only where at most half the groups read halo columns; else format 3 throughout)
            assert st["paired_rows"] in (1, 2), st
            report["paired_rows_fmt4"], report["tiled_planes"] = int(st["paired_rows"]), int(st["tiled_planes"])
        mat.set_halo(plan.nbr_rank, plan.send_ptr, plan.send_idx, plan.recv_ptr)
        assert 0 < st["n_interior_slices"] <= st["n_slices"]
        op = api.HipStencilOperator(mat, -1.0, 0.0)
        # SpMV through the halo exchange
        xv = api.DeviceVector.from_numpy(ctx, x_glob[gid], n_halo=loc.n_halo)
        yv = api.DeviceVector(ctx, n, loc.n_halo)
        for _ in range(2):
            mat.apply(-1.0, 0.0, xv, yv)
        err = np.abs(yv.to_numpy() - y_ref[gid]).max() / np.abs(y_ref).max()
        assert err <= 1e-13, ("spmv", fmt, err)
        # global reductions
        d = api.dot_product(xv, yv)
        d_ref = float(x_glob @ y_ref)
        assert abs(d - d_ref) <= 1e-11 * abs(d_ref), (d, d_ref)
        nrm = api.norm_2(xv)
        assert abs(nrm - np.linalg.norm(x_glob)) <= 1e-12 * nrm
        # device-resident solver loops
        b = api.DeviceVector(ctx, n, loc.n_halo)
        api.fill_with(b, 1.0)
        for cls, kind, kw in ((api.CgSolver, "cg", {}), (api.BiCgStabSolver, "bicgstab", {}),
                              (api.GmresSolver, "gmres", {"num_inner_iterations": 20, "gram_schmidt": 0}),
                              (api.GmresSolver, "gmres", {"num_inner_iterations": 20, "gram_schmidt": 1})):
            x = api.DeviceVector(ctx, n, loc.n_halo)
            s = cls()
            for k, v in kw.items():
                setattr(s, k, v)
            assert s.solve(x, b, op), (kind, fmt)
            ref = oracle.solve(kind, ref_op, np.ones(glob.n_cells), num_inner_iterations=kw.get("num_inner_iterations", 50))
            assert ref.converged
            # (BiCGStab's count is a draw among roundings -- NOTES.md 5c.  On the 64 x 16 x 24 box the ORACLE itself gives 54
            #  (strict build), 62 (the same source with FMA contraction) and 53 ... 63 over sixteen runs whose applies are
            #  perturbed by one unit in the last place (oracle.GatherOperator seeds 1-8, strict / devlike); the device: 59
            #  with the backend's free contraction, 64 with -ffp-contract=on.  The solution parity below is the check.)
            tol = 0.2 if kind == "bicgstab" else 0.05
            assert abs(s.iteration - ref.iterations) <= max(2, int(tol * ref.iterations)), (kind, s.iteration, ref.iterations)
            # every rank must have taken the same decision
            its = np.array([float(s.iteration)])
            lo, hi = its.copy(), its.copy()
            import torch

            td.all_reduce(torch.from_numpy(lo), op=td.ReduceOp.MIN)
            td.all_reduce(torch.from_numpy(hi), op=td.ReduceOp.MAX)
            assert lo[0] == hi[0] == s.iteration
            rel = np.linalg.norm(x.to_numpy() - ref.x[gid]) / np.linalg.norm(ref.x[gid])
            # (two BiCGStab solves that stop at different iterations agree to the solve tolerance only)
            # (... as do two GMRES solves whose last restart cycle ends differently: tests/test_gpu_parity.py uses 5e-6 too)
            assert rel <= (5e-6 if kind == "gmres" else 2e-6 if kind == "bicgstab" else 1e-7), (kind, fmt, rel)
            report[f"{kind}{kw.get('gram_schmidt', '')}_fmt{fmt}"] = [int(s.iteration), int(ref.iterations), float(rel)]
        # a statement-level solver on the same transport (host loop: dots through the all-reduce)
        x = api.DeviceVector(ctx, n, loc.n_halo)
        s = api.CgsSolver()
        assert s.solve(x, b, op)
        ref = oracle.solve("cgs", ref_op, np.ones(glob.n_cells))
        assert abs(s.iteration - ref.iterations) <= max(2, int(0.1 * ref.iterations))
        assert np.linalg.norm(x.to_numpy() - ref.x[gid]) <= 1e-6 * np.linalg.norm(ref.x[gid])
        mat.close()
    ctx.sync()
    td.barrier()  # nobody unmaps / frees a peer window another rank's kernels may still write to
    ctx.close()
    td.barrier()
    with open(os.path.join(os.environ["STORM_REPORT_DIR"], f"rank{rank}.json"), "w") as f:
        json.dump(report, f)
    td.destroy_process_group()


if __name__ == "__main__":
    main()
