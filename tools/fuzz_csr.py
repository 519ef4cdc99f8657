#!/usr/bin/env python3
"""Random CSR operators -- 1 ... 6 000 rows (1, 63, 64, 65 ... among them), empty to 40 entries per row, a few rows of up to
400 entries (the CSR tail), empty rows, three distinct values (the dictionary formats) -- through every record format
(option spmv_dict 0 ... 4) and ELL caps, y = beta x + alpha A x against scipy to the rounding bound of the records'
difference form.  python tools/fuzz_csr.py [seed] [cases]"""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(seed=0, cases=300, verbose=True):
    from stormruler_amd import api

    rng = np.random.default_rng(seed)
    ctx = api.Context(0)
    bad = 0
    for case in range(cases):
        bad += _one(api, ctx, rng, case, verbose)
    ctx.close()
    return bad


def _one(api, ctx, rng, case, verbose):
    bad = 0
    if True:
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 129, 1000, 4099, int(rng.integers(1, 6000))]))
        dens = float(rng.choice([0.0, 0.5 / max(n, 1), 3.0 / max(n, 1), 8.0 / max(n, 1), 40.0 / max(n, 1)]))
        a = sp.random(n, n, density=min(1.0, dens), random_state=int(rng.integers(1 << 30)), format="lil", data_rvs=lambda k: rng.standard_normal(k))
        if n > 10 and rng.random() < 0.4:  # a few long rows (CSR tail) and an empty one
            for r in rng.integers(0, n, 3):
                cols = rng.choice(n, size=min(n, int(rng.integers(20, 400))), replace=False)
                a[r, cols] = rng.standard_normal(len(cols))
            a[int(rng.integers(0, n)), :] = 0
        if rng.random() < 0.3:  # few distinct values: the dictionary formats
            a = a.tocsr(); a.data = rng.choice([0.25, -1.0, 3.5], size=a.data.shape); a = a.tolil()
        a = a.tocsr(); a.eliminate_zeros(); a.sort_indices()
        fmt = int(rng.choice([0, 1, 2, 3, 4]))
        cap = int(rng.choice([0, 4, 8, 32]))
        ctx.set_option("spmv_dict", fmt)
        if cap: ctx.set_option("ell_cap", cap)
        try:
            mat = api.StencilMatrix.from_csr(ctx, a)
        finally:
            ctx.set_option("spmv_dict", 4); ctx.set_option("ell_cap", 0) if cap else None
        x = rng.standard_normal(n)
        alpha, beta = float(rng.standard_normal()), float(rng.choice([0.0, 1.0, rng.standard_normal()]))
        xd, yd = api.DeviceVector.from_numpy(ctx, x), api.DeviceVector(ctx, n)
        mat.apply(alpha, beta, xd, yd)
        want = beta * x + alpha * (a @ x)
        got = yd.to_numpy()
        rowabs = np.asarray(np.abs(a).sum(axis=1)).ravel()
        # (the records hold a row as sum w (x_col - x_i) + ext x_i: the bound carries |x_i| sum |w| too)
        scale = np.abs(alpha) * (np.abs(a) @ np.abs(x) + rowabs * np.abs(x)) * max(1.0, a.getnnz(axis=1).max() / 16.0 if n and a.nnz else 1.0) + np.abs(beta * x) + 1e-300
        err = np.max(np.abs(got - want) / scale) if n else 0.0
        if not (err <= 1e-13):
            bad += 1
            if verbose:
                print("MISMATCH", case, n, dens, fmt, cap, err, mat.stats().get("tail_rows"))
        mat.close()
    return bad


if __name__ == "__main__":
    n_bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 300)
    print("bad", n_bad)
    sys.exit(1 if n_bad else 0)
