#!/usr/bin/env python3
"""How far does BiCGStab's iteration count on the 256^3 Poisson problem (BASELINE config 3's per-GPU block) move when
NOTHING changes but last-place roundings?  CPU only (the oracle's investigation builds), ~7 processes, ~1 h:

    python tools/bicgstab_draw_study.py            # writes the `perturbation_study` block of
                                                   # tests/golden/full_size_bicgstab256.json

The reference-order oracle needs 355 iterations (the fixture's headline figure); the same source with other summation
orders 350 / 352 / 361 (`summation_order_study`); the HIP loops 378.  The HIP SpMV rounds differently from the
reference's face loop (weights divided once, rows gathered in face order, difference form, FMAs), so this tool runs
the oracle's BiCGStab -- the reference's statements, SolverBiCgStab.hpp:93-165 -- on that operator form
(`oracle_gather_apply`) and perturbs every apply by at most one unit in the last place, seeded:

  * family "devlike": tree-shaped sums + FMA contraction in the vector updates (liboracle_devlike.so: the device's
    arithmetic on the CPU), seeds 0 (unperturbed) .. 31;
  * family "strict":  the reference's sequential sums, no contraction (liboracle.so), seeds 1 .. 16 -- the spread is
    there with the reference's own summation order too.

`tests/test_gpu_full_size.py` asserts that the device's count lies within three sample standard deviations of the mean
of these 48 counts: the bound comes from this committed data, not from a percentage.  (`--resume`: keep the runs the
fixture already holds and add the missing ones.)"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHM = "/dev/shm/storm_bicgstab_study"
RUNS = [("devlike", s) for s in range(32)] + [("strict", s) for s in range(1, 17)]


def build_arrays(n):
    from oracle import oracle
    from stormruler_amd import mesh

    g = mesh.structured_box(n)
    op = oracle.GatherOperator(g, -1.0, 0.0)
    os.makedirs(SHM, exist_ok=True)
    np.save(os.path.join(SHM, "col.npy"), op.col)
    np.save(os.path.join(SHM, "w.npy"), op.w)
    np.save(os.path.join(SHM, "ext.npy"), op.ext)


def run_one(n, family, seed, out):
    import ctypes as C

    from oracle import oracle

    col = np.load(os.path.join(SHM, "col.npy"), mmap_mode="r")
    w = np.load(os.path.join(SHM, "w.npy"), mmap_mode="r")
    ext = np.load(os.path.join(SHM, "ext.npy"), mmap_mode="r")
    N, W = col.shape
    L = oracle.lib(family)
    c = oracle._GatherOp(N, W, col.ctypes.data_as(oracle.i64p), w.ctypes.data_as(oracle.f64p),
                         ext.ctypes.data_as(oracle.f64p), -1.0, 0.0, seed, 0)

    class Op:
        fn = C.cast(L.oracle_gather_apply, C.c_void_p)
        ctx = C.cast(C.pointer(c), C.c_void_p)

    t = time.time()
    r = oracle.solve("bicgstab", Op, np.ones(N), variant=family)
    json.dump({"family": family, "seed": seed, "edge": n, "iterations": int(r.iterations), "converged": bool(r.converged),
               "relative_error": r.relative_error, "x_norm2": float(np.sqrt(np.sum(r.x * r.x))),
               "history_head": [float(v) for v in r.history[:24]], "seconds": time.time() - t}, open(out, "w"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edge", type=int, default=256)
    ap.add_argument("--jobs", type=int, default=7)
    ap.add_argument("--one", nargs=3, metavar=("FAMILY", "SEED", "OUT"))
    ap.add_argument("--resume", action="store_true", help="keep the runs already in the fixture, add the missing ones")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "full_size_bicgstab256.json"))
    a = ap.parse_args()
    if a.one:
        run_one(a.edge, a.one[0], int(a.one[1]), a.one[2])
        return
    kept = []
    if a.resume and a.edge == 256:
        kept = json.load(open(a.out)).get("perturbation_study", {}).get("runs", [])
    have = {(r["family"], r["seed"]) for r in kept}
    build_arrays(a.edge)
    try:
        jobs, running, results = [j for j in RUNS if j not in have], [], list(kept)
        while jobs or running:
            while jobs and len(running) < a.jobs:
                fam, seed = jobs.pop(0)
                out = os.path.join(SHM, f"{fam}_{seed}.json")
                running.append((out, subprocess.Popen([sys.executable, __file__, "--edge", str(a.edge), "--one", fam, str(seed), out],
                                                      env=dict(os.environ, OMP_NUM_THREADS="1"))))
            time.sleep(5)
            for job in list(running):
                if job[1].poll() is not None:
                    assert job[1].returncode == 0, job[0]
                    results.append(json.load(open(job[0])))
                    print(results[-1]["family"], results[-1]["seed"], results[-1]["iterations"], f"{results[-1]['seconds']:.0f} s", flush=True)
                    running.remove(job)
    finally:
        for f in os.listdir(SHM):
            os.remove(os.path.join(SHM, f))
        os.rmdir(SHM)
    results.sort(key=lambda r: (r["family"], r["seed"]))
    counts = [r["iterations"] for r in results]
    mean = float(np.mean(counts))
    std = float(np.std(counts, ddof=1))
    block = {"generator": "tools/bicgstab_draw_study.py", "edge": a.edge, "mean_iterations": mean, "std_iterations": std,
             "what": "oracle BiCGStab (SolverBiCgStab.hpp:93-165) on the operator in the HIP kernels' arithmetic form "
                     "(oracle_gather_apply), every apply perturbed by <= 1 ulp (seed 0: unperturbed); family devlike = tree "
                     "sums + FMA contraction, family strict = the reference's sequential sums without contraction",
             "runs": results, "min_iterations": min(counts), "max_iterations": max(counts)}
    if a.edge == 256:
        fx = json.load(open(a.out))
        fx["perturbation_study"] = block
        with open(a.out, "w") as f:
            json.dump(fx, f, indent=0)
    else:
        print(json.dumps(block)[:2000])
    print("iterations:", sorted(counts))


if __name__ == "__main__":
    main()
