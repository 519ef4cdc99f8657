#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel and counter, the number of dispatches and the mean value.

    python tools/pmc_summary.py <dir with *_counter_collection.csv> [...]

Also writes profiles/spmv_hbm_traffic.json when --traffic-json is given: FETCH_SIZE (KiB; doubled, the
gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md) + WRITE_SIZE (KiB) of the SpMV kernel with the
fused-dot epilogue, per launch."""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def short(name: str) -> str:
    return re.sub(r"\(.*", "", name).strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--traffic-json", default=None)
    ap.add_argument("--record-format", default=None)
    ap.add_argument("--algorithmic-bytes", type=float, default=None)
    ap.add_argument("--format-bytes", type=float, default=None)
    args = ap.parse_args()
    acc = defaultdict(lambda: [0, 0.0])
    for d in args.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = (row["Counter_Name"], short(row["Kernel_Name"]))
                    acc[k][0] += 1
                    acc[k][1] += float(row["Counter_Value"])
    lines = []
    for (counter, kern), (n, tot) in sorted(acc.items()):
        lines.append(f"{counter} | {kern} | n={n} | avg={tot / n:.1f}")
    print("\n".join(lines))
    if args.traffic_json:
        def avg(counter, pred):
            sel = [(n, t) for (c, k), (n, t) in acc.items() if c == counter and pred(k)]
            n = sum(a for a, _ in sel)
            return sum(t for _, t in sel) / n if n else None

        # the launches of a CG solve: the fused-dot instantiation <true, ...> of the SpMV kernels; the byte-indexed
        # formats (headline operator) and the fp64 records (roofline_general) are reported separately
        is_step = lambda k: ("spmv_canon_tile_kernel<true" in k and ", true>" in k) or "cg_step_march_kernel" in k  # noqa: E731  (the fused CG step)
        is_fmt = lambda k: (("spmv_canon_kernel<true" in k) or ("spmv_pair_kernel<true" in k) or ("spmv_dict_kernel<true" in k) or  # noqa: E731
                            ("spmv_canon_tile_kernel<true" in k and ", true>" not in k))
        is_sell = lambda k: "spmv_sell_kernel<true, true" in k  # noqa: E731
        method = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950)"
        out = {}
        fetch, write = avg("FETCH_SIZE", is_step), avg("WRITE_SIZE", is_step)
        if fetch is None or write is None:  # no fused CG step in the run: the SpMV with the fused dot
            fetch, write = avg("FETCH_SIZE", is_fmt), avg("WRITE_SIZE", is_fmt)
        if fetch is None or write is None:  # the operator did not qualify for a byte-indexed format
            fetch, write = avg("FETCH_SIZE", is_sell), avg("WRITE_SIZE", is_sell)
        if fetch is not None and write is not None:
            traffic = (2.0 * fetch + write) * 1024.0
            out = {"traffic_bytes_per_launch": traffic, "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB": write,
                   "record_format": args.record_format, "method": method}
            if args.format_bytes:
                out["ratio_to_format_bytes"] = traffic / args.format_bytes
            if args.algorithmic_bytes:
                out["ratio_to_algorithmic_bytes"] = traffic / args.algorithmic_bytes
        fetch, write = avg("FETCH_SIZE", is_sell), avg("WRITE_SIZE", is_sell)
        if fetch is not None and write is not None:
            traffic = (2.0 * fetch + write) * 1024.0
            out["general"] = {"traffic_bytes_per_launch": traffic, "FETCH_SIZE_KiB_raw": fetch,
                              "WRITE_SIZE_KiB": write, "record_format": "fp64 weights + int32 columns",
                              "method": method}
            if args.algorithmic_bytes:
                out["general"]["ratio_to_algorithmic_bytes"] = traffic / args.algorithmic_bytes
        # the stand-alone applies of bench.py's `spmv` block (no fused dot)
        # (the tiled kernel where it ran: bench.py's event-pair floor launches spmv_canon_kernel<false on a tiny operator)
        plain_names = ("spmv_canon_tile_kernel<false",) if any("spmv_canon_tile_kernel<false" in k for _, k in acc) else (
            "spmv_canon_kernel<false", "spmv_pair_kernel<false", "spmv_dict_kernel<false")
        is_plain_lat = lambda k: any(t in k for t in plain_names)  # noqa: E731
        is_plain_sell = lambda k: "spmv_sell_kernel<true, false" in k  # noqa: E731
        for name, pred in (("spmv_alone_lattice", is_plain_lat), ("spmv_alone_general", is_plain_sell)):
            fetch, write = avg("FETCH_SIZE", pred), avg("WRITE_SIZE", pred)
            if fetch is not None and write is not None:
                out[name] = {"traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0, "FETCH_SIZE_KiB_raw": fetch,
                             "WRITE_SIZE_KiB": write, "method": method}
        if out:
            with open(args.traffic_json, "w") as fh:
                json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
