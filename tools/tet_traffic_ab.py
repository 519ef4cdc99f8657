#!/usr/bin/env python3
"""HBM traffic of the fp64-record SpMV on the tetrahedral box (12.6 M cells, Morton order) under block -> XCD mappings and
cell orderings: VERDICT r05 item 8's bounded experiment (1.19 x the algorithmic bytes: x re-fetched).

    python tools/tet_traffic_ab.py child <ordering> <xcd_group>     the process a profiler wraps: 30 rotating applies
    python tools/tet_traffic_ab.py                                  the sweep: per setting two rocprofv3 --pmc passes
                                                                    (FETCH_SIZE, WRITE_SIZE; FETCH_SIZE doubled: gfx950)
                                                                    + an unprofiled timing run; one JSON line each

Every byte a CU asks another level for is counted where it leaves the XCD's L2 (FETCH_SIZE): a row of x that two XCDs
gather is fetched twice -- into two L2s -- whether HBM or the Infinity Cache serves it.  spmv_xcd_remap = G deals runs of G
consecutive blocks (256 G rows) to one XCD."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(ordering, group, n3=128):
    import numpy as np

    from stormruler_amd import api, host_mesh, io_tetgen

    pos, bf, cells = io_tetgen.tet_box(n3)
    hm = host_mesh.HostMesh.from_simplices(pos, bf, np.ones(len(bf), np.int64), cells)
    if ordering != "file":
        hm.order_cells(ordering)
    ctx = api.Context(0)
    ctx.set_option("spmv_xcd_remap", group)
    mat = hm.create_operator(ctx)
    n = mat.stats()["n_rows"]
    xs = [api.DeviceVector.from_numpy(ctx, np.sin(0.37 * np.arange(n))) for _ in range(3)]
    ys = [api.DeviceVector(ctx, n) for _ in range(3)]
    for i in range(6):
        mat.apply(-1.0, 0.0, xs[i % 3], ys[i % 3])
    ctx.set_option("profile_spmv", 1)
    for i in range(30):
        mat.apply(-1.0, 0.0, xs[i % 3], ys[i % 3])
    smp = ctx.spmv_profile_samples()
    st = mat.stats()
    print(json.dumps({"median_ms": float(np.median(smp)), "rows": n, "nnz_offdiag": st["nnz_offdiag"]}), flush=True)
    mat.close()
    ctx.close()


def sweep():
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    settings = [("morton", 8), ("morton", 0), ("morton", 32), ("morton", 128), ("morton", 1), ("hilbert", 8), ("hilbert", 64)]
    for ordering, group in settings:
        out = {"ordering": ordering, "spmv_xcd_remap": group}
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", ordering, str(group)], capture_output=True, text=True, cwd=ROOT)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            print(json.dumps(dict(out, error=(p.stdout + p.stderr)[-300:])), flush=True)
            continue
        t = json.loads(line[-1])
        alg = 24 * t["rows"] + 12 * t["nnz_offdiag"]
        out.update(median_ms=t["median_ms"], frac_8d=alg / (t["median_ms"] * 1e-3) / 1e9 / 8000.0, algorithmic_bytes_8d=alg)
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="storm_tet_pmc_", dir="/tmp")
            try:
                q = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                    os.path.abspath(__file__), "child", ordering, str(group)], capture_output=True, text=True, cwd="/tmp",
                                   env=dict(os.environ, TMPDIR="/tmp"), timeout=600)
                n_, tot = 0, 0.0
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    with open(f, newline="") as fh:
                        for row in csv.DictReader(fh):
                            if row["Counter_Name"] == counter and "spmv_sell_kernel<true, false" in row["Kernel_Name"]:
                                n_ += 1
                                tot += float(row["Counter_Value"])
                out[counter + "_KiB_per_launch"] = tot / n_ if n_ else None
                out[counter + "_launches"] = n_
                if q.returncode != 0:
                    out[counter + "_rc"] = q.returncode
            finally:
                shutil.rmtree(d, ignore_errors=True)
        f_, w_ = out.get("FETCH_SIZE_KiB_per_launch"), out.get("WRITE_SIZE_KiB_per_launch")
        if f_ and w_:
            out["traffic_bytes_per_launch"] = (2.0 * f_ + w_) * 1024.0
            out["traffic_over_8d_bytes"] = out["traffic_bytes_per_launch"] / alg
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        sweep()
