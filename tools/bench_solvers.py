#!/usr/bin/env python3
"""Secondary measurements (not the headline line): iteration rates of the other BASELINE configs on
one GPU -- BiCGStab on 256^3 Poisson (config 3's per-GPU block), GMRES(30) on 128^3
convection-diffusion (config 4), CG on 64^3/128^3 -- with fixed iteration counts (tolerances off)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api, mesh  # noqa: E402


def timed(ctx, make_solver, x_factory, b, op, iters):
    for it in (max(2, iters // 10), iters):
        s = make_solver()
        s.num_iterations = it
        s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
        x = x_factory()
        ctx.sync()
        t = time.perf_counter()
        s.solve(x, b, op)
        ctx.sync()
        dt = time.perf_counter() - t
    return iters / dt, dt / iters * 1e3, s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="cg64,cg128,cg256,bicgstab256,gmres128cd,gmres128cd_cgs2")
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    ctx = api.Context(0)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    cache = {}

    def poisson(n):
        if n not in cache:
            g = mesh.structured_box(n)
            cache[n] = (g, api.StencilMatrix.from_face_graph(ctx, g))
        return cache[n]

    for name in args.which.split(","):
        if name.startswith("cg") or name.startswith("bicgstab"):
            kind = "cg" if name.startswith("cg") else "bicgstab"
            n = int(name[len(kind):])
            g, mat = poisson(n)
            op = api.HipStencilOperator(mat, -1.0, 0.0)
            cls = api.CgSolver if kind == "cg" else api.BiCgStabSolver
            iters = 200 if n >= 128 else 500
            mk = cls
            N = g.n_cells
            nnz = mat.stats()["nnz_offdiag"]
            bspmv = 24 * N + 12 * nnz
            alg = bspmv + 96 * N if kind == "cg" else 2 * bspmv + 192 * N
        else:
            n = 128
            g = mesh.structured_box(n)
            wi, wo, de = mesh.convection_diffusion_weights(g, 1e-2, (1.0, 0.5, 0.25))
            mat = api.StencilMatrix.from_face_weights(ctx, g.n_cells, 0, g.inner, g.outer, wi, wo, de)
            op = api.HipStencilOperator(mat, 1.0, 0.0)
            gs = 1 if name.endswith("cgs2") else 0

            def mk(gs=gs):
                s = api.GmresSolver()
                s.num_inner_iterations, s.gram_schmidt = 30, gs
                return s

            iters = 150
            N = g.n_cells
            bspmv = 24 * N + 12 * mat.stats()["nnz_offdiag"]
            # reference MGS op list averaged over a restart cycle: k+1 (dot 16N + axpy 24N) + norm/scale 24N
            alg = bspmv + np.mean([(k + 1) * 40 * N + 24 * N for k in range(30)])
        b = api.DeviceVector(ctx, N)
        api.fill_with(b, 1.0)
        rate, ms, s = timed(ctx, mk, lambda: api.DeviceVector(ctx, N), b, op, iters)
        print(json.dumps({"case": name, "cells": N, "iterations_per_sec": rate, "ms_per_iteration": ms,
                          "reference_op_list_bytes_per_iteration": float(alg),
                          "reference_equivalent_GBs": float(alg) * rate / 1e9}))


if __name__ == "__main__":
    main()
