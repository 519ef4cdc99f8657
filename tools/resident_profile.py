"""Where a resident-path CG iteration spends its time (csrc/resident.hip, option resident_profile): the blocks time the
phases of their loop with the 100 MHz counter; mean and max over blocks, microseconds per iteration.

    python tools/resident_profile.py [--shapes 64,128] [--iters 300]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402

PHASES = {"cg": ["halo_of_new_direction", "apply", "pz_partials", "allreduce_1", "r_update", "allreduce_2", "x_p_update_publish"],
          "bicgstab": ["p_update_publish", "halo_of_p", "apply_1", "rtv_allreduce", "s_update_publish_halo", "apply_2",
                       "ts_tt_allreduce", "x_r_update_allreduce"]}


# (BiCGStab with --early 1, boxes of at most six planes -- res_bicgstab_early_kernel: "p_update_publish" is the update of p on
#  the own rows AND on the halo (nothing is published there), "halo_of_p" is empty, "rtv_allreduce" includes the surface of v
#  going out and coming in, "s_update_publish_halo" is the update of s on own rows and halo, "x_r_update_allreduce" includes
#  the surface of the new residual going out and coming in.)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="64,128")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--bicgstab", action="store_true")
    ap.add_argument("--early", type=int, default=0, help="option resident_early (CG: the exchange under the second all-reduce)")
    args = ap.parse_args()
    ctx = api.Context(0)
    ctx.set_option("resident_profile", 1)
    ctx.set_option("resident_early", args.early)
    for tok in args.shapes.split(","):
        dims = [int(v) for v in tok.split("x")]
        g = mesh.structured_box(*dims)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        op = api.HipStencilOperator(mat, -1.0, 0.0)
        b = api.DeviceVector.from_numpy(ctx, 1.0 + 0.25 * np.sin(0.01 * np.arange(g.n_cells)))
        x = api.DeviceVector(ctx, g.n_cells)
        s = api.BiCgStabSolver() if args.bicgstab else api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = args.iters, 0.0, 0.0
        s.solve(x, b, op)
        rec = {"shape": dims, "iterations": int(s.iteration)}
        for kind in ("mean", "max"):
            rec[kind + "_us_per_iteration"] = {
                name: round(ctx.counter(f"resident_phase_{kind}_{k}") * 0.01 / args.iters, 2) for k, name in enumerate(PHASES["bicgstab" if args.bicgstab else "cg"])}
        rec["sum_of_means"] = round(sum(rec["mean_us_per_iteration"].values()), 2)
        print(json.dumps(rec), flush=True)
        mat.close()
    ctx.close()


if __name__ == "__main__":
    main()
