"""BASELINE config 5's time step on one GPU: the lid-driven cavity at 128^3 (stormruler_amd/cavity.py), seconds per
step, CG iterations per step and microseconds per CG iteration, with the path the pressure solves took.

    python tools/cavity_rate.py [--edge 128] [--steps 6] [--resident 0|1]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, cavity  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=128)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--resident", type=int, default=1)
    args = ap.parse_args()
    ctx = api.Context(0)
    ctx.set_option("resident_path", args.resident)
    dev = cavity.CavityProjection(ctx, args.edge, nu=0.01)
    its, secs = [], []
    r0 = ctx.counter("resident_solves")
    for _ in range(args.steps):
        it, sec, ok = dev.step()
        its.append(int(it)), secs.append(sec)
    ctx.sync()
    # the solve alone, warm-started like the step's: re-solve the last system from p = 0 with the step's tolerance
    print(json.dumps({"edge": args.edge, "resident_path": args.resident, "cg_iterations_per_step": its,
                      "seconds_per_step": [round(s, 5) for s in secs],
                      "resident_solves": ctx.counter("resident_solves") - r0,
                      "ms_per_step_last": round(1e3 * secs[-1], 3),
                      "us_per_cg_iteration_upper_bound": round(1e6 * secs[-1] / max(its[-1], 1), 2)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
