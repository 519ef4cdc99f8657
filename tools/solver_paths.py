#!/usr/bin/env python3
"""Iteration rates of the solver PATHS of this library on one MI355X (one JSON line per case, for profiles/):

    fused          stencil operator, the fused kernels of csrc/solvers.hip (CG / BiCGStab / GMRES only)
    engine         stencil operator through the general engine of csrc/krylov.hip (`generic_solvers = 1`)
    engine-lambda  the operator handed over as a lambda through make_operator -- what the reference's only call
                   site does (Playground.cpp:151-167) -- device-resident loop, the callback only enqueues
    engine-jacobi  stencil operator + the diagonal preconditioner behind pre_op
    stepping       the same lambda, `device_loop = False`: the host runs the reference's loop over
                   init / iterate (one host wait per iteration)

on (a) the 256^3 Poisson box and (b) the reference's own Triangle mesh step.1 (79 672 cells; operator
x - 1e-2 div grad x as in BASELINE.md 2).  Fixed iteration counts, tolerances off.

    python tools/solver_paths.py [--meshes box256,step1] [--solvers cg,bicgstab,...]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stormruler_amd import api, io_tetgen, mesh  # noqa: E402

SOLVERS = {"cg": api.CgSolver, "bicgstab": api.BiCgStabSolver, "gmres30": api.GmresSolver, "cgs": api.CgsSolver,
           "tfqmr": api.TfqmrSolver, "tfqmr1": api.Tfqmr1Solver, "bicgstabl2": api.BiCgStabLSolver,
           "idrs4": api.IdrsSolver}


def make(kind):
    s = SOLVERS[kind]()
    if kind == "gmres30":
        s.num_inner_iterations = 30
    s.absolute_error_tolerance = s.relative_error_tolerance = 0.0
    return s


def rate(ctx, kind, op, b, n, iters, generic=False, device_loop=True, pre=None):
    out = None
    for rep, it in enumerate((max(4, iters // 8), iters, iters, iters)):  # first pass = warm-up; best of three
        s = make(kind)
        s.num_iterations = it
        s.device_loop = device_loop
        s.pre_op = pre() if pre else None
        x = api.DeviceVector(ctx, n)
        ctx.set_option("generic_solvers", int(generic))
        ctx.sync()
        t = time.perf_counter()
        s.solve(x, b, op)
        ctx.sync()
        dt = time.perf_counter() - t
        ctx.set_option("generic_solvers", 0)
        assert s.iteration == it and np.isfinite(s.absolute_error)
        if rep > 0 and (out is None or dt / it * 1e6 < out["us_per_iter"]):
            out = {"iter_per_s": it / dt, "us_per_iter": dt / it * 1e6, "iterations": it}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", default="box256,step1,box64")
    ap.add_argument("--solvers", default=",".join(SOLVERS))
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (repeatable)")
    a = ap.parse_args()
    ctx = api.Context(0)
    for kv in a.opt:
        key, value = kv.split("=")
        ctx.set_option(key, int(value))
    for name in a.meshes.split(","):
        if name.startswith("box"):
            n = int(name[3:])
            g = mesh.structured_box(n)
            alpha, beta, iters = -1.0, 0.0, (200 if n >= 128 else 400)
            b_host = np.ones(g.n_cells)
        else:
            g = io_tetgen.read_triangle(os.path.join(ROOT, "tests", "golden", "mesh", "step.1."))
            g = mesh.FaceGraph(g.n_cells, 2, g.inner, g.outer, g.area, g.center, g.volume, b_center=np.zeros((0, 2)))
            alpha, beta, iters = -1.0e-2, 1.0, 400
            b_host = np.sin(3 * g.center[:, 0]) * np.cos(7 * g.center[:, 1])
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        op = api.HipStencilOperator(mat, alpha, beta)
        lam = api.make_operator(lambda y, x: mat.apply(alpha, beta, x, y))
        b = api.DeviceVector.from_numpy(ctx, b_host)
        for kind in a.solvers.split(","):
            row = {"mesh": name, "cells": g.n_cells, "solver": kind, "record_format": mat.stats()["paired_rows"] and "paired"
                   or (mat.stats()["value_dictionary_size"] and "dictionary") or "fp64"}
            if kind in ("cg", "bicgstab", "gmres30"):
                row["fused"] = rate(ctx, kind, op, b, g.n_cells, iters)
            row["engine"] = rate(ctx, kind, op, b, g.n_cells, iters, generic=True)
            row["engine-lambda"] = rate(ctx, kind, lam, b, g.n_cells, iters)
            row["engine-jacobi"] = rate(ctx, kind, op, b, g.n_cells, iters, generic=True, pre=api.JacobiPreconditioner)
            row["stepping"] = rate(ctx, kind, lam, b, g.n_cells, min(iters, 200), device_loop=False)
            print(json.dumps(row), flush=True)
        mat.close()


if __name__ == "__main__":
    main()
