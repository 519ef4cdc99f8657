"""Pressure-Poisson time-step driver: lid-driven cavity by Chorin projection (BASELINE config 5).

This is the *caller* of the hot path, shaped like the reference's time loop
(source_apps/playground/Playground.cpp:133-210): build the operators once, then per step form a
right-hand side, warm-start the unknown from the previous step (``c_hat <<= c``, :150) and call
``solve<CgSolver>``, timing each step (:186-206).  The reference itself has no incompressible
Navier-Stokes code at this commit (its README only announces it), so the scheme is this build's own:

    u*_d = u_d + dt ( nu L_D u_d + nu s_d - sum_e u_e .* (G_e u_d) )        d, e in {x, y, z}
    L_N p = (1/dt) sum_d D_d u*_d            (CG, pure Neumann, warm start)
    u_d  = u*_d - dt G_d p

Every linear piece is a face-graph operator applied by the SpMV kernel: ``L_D`` the Dirichlet
diffusion stencil, ``L_N`` the Neumann one (no wall faces), ``G_e`` / ``D_e`` central face-flux
gradient / divergence written as face weights (``gradient_weights``); the only non-SpMV work is the
elementwise product ``u_e .* (G_e u_d)`` (``storm_hip_vmul_add``).  All state stays in HBM between
steps; one step is 18 SpMVs + 1 CG solve.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .mesh import FaceGraph, face_coefficients, structured_box

__all__ = ["gradient_weights", "lid_source", "CavityOperators", "build_cavity_operators", "CavityProjection"]


def gradient_weights(g: FaceGraph, axis: int, wall_value_zero: bool) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Face weights of the central (Green-Gauss) derivative along ``axis``.

    ``(G phi)_i = (1/V_i) sum_f s_if A_f n_f,axis phi_f`` with ``phi_f`` the face average and ``n_f`` the
    unit vector inner -> outer.  ``wall_value_zero = False`` extrapolates ``phi`` to the wall (pressure
    gradient); ``True`` uses a zero wall value (divergence of a velocity that vanishes at / slides along
    the walls).  Returns ``(w_inner, w_outer, diag_extra)`` for ``StencilMatrix.from_face_weights``.
    """
    coef, b_coef = face_coefficients(g)
    d = g.center[g.outer] - g.center[g.inner]
    n_ax = d[:, axis] / (g.area / coef)
    w_inner = g.area * n_ax / (2.0 * g.volume[g.inner])
    w_outer = -g.area * n_ax / (2.0 * g.volume[g.outer])
    diag = np.zeros(g.n_total)
    np.add.at(diag, g.inner, g.area * n_ax / g.volume[g.inner])
    np.add.at(diag, g.outer, -g.area * n_ax / g.volume[g.outer])
    if g.n_bfaces and not wall_value_zero:
        db = g.b_center - g.center[g.b_cell]
        nb = db[:, axis] / (g.b_area / b_coef)
        np.add.at(diag, g.b_cell, g.b_area * nb / g.volume[g.b_cell])
    return w_inner, w_outer, diag[: g.n_cells].copy()


def lid_source(g: FaceGraph, lid_axis: int = 2, u_lid: float = 1.0, lid_coord: float = None) -> np.ndarray:
    """Diffusive flux from the moving lid (the wall at ``lid_coord`` along ``lid_axis``; default: the
    largest wall coordinate of ``g``) into the tangential velocity: ``(A_b / d_b) u_lid / V_i`` on lid
    cells (the Dirichlet ghost value of the lid faces).  On a partitioned mesh pass the global
    ``lid_coord``: only the ranks that own lid cells get a non-zero source."""
    _, b_coef = face_coefficients(g)
    if g.n_bfaces == 0:
        return np.zeros(g.n_cells)
    if lid_coord is None:
        lid_coord = g.b_center[:, lid_axis].max()
    top = np.isclose(g.b_center[:, lid_axis], lid_coord)
    s = np.zeros(g.n_cells)
    np.add.at(s, g.b_cell[top], b_coef[top] * u_lid / g.volume[g.b_cell[top]])
    return s


@dataclass
class CavityOperators:
    """Host description of every operator of the scheme (shared by the device driver and the CPU check)."""

    g: FaceGraph
    g_neumann: FaceGraph
    grad: List[Tuple[np.ndarray, np.ndarray, np.ndarray]]  # G_e weights
    div: List[Tuple[np.ndarray, np.ndarray, np.ndarray]]   # D_e weights
    lid: np.ndarray


def build_cavity_operators(n: int, graph: FaceGraph = None, lid_coord: float = 1.0) -> CavityOperators:
    """Operators of the unit-cube cavity; ``graph`` = a rank's local face graph of that cube (owned +
    halo cells, e.g. from ``partition.slab_partition``/``partition_graph``) for a partitioned run."""
    g = structured_box(n) if graph is None else graph
    gn = FaceGraph(g.n_cells, g.dim, g.inner, g.outer, g.area, g.center, g.volume,
                   b_center=np.zeros((0, g.dim)), n_halo=g.n_halo, global_id=g.global_id,
                   halo_owner=g.halo_owner)
    grad = [gradient_weights(g, e, wall_value_zero=False) for e in range(3)]
    div = [gradient_weights(g, e, wall_value_zero=True) for e in range(3)]
    return CavityOperators(g, gn, grad, div, lid_source(g, lid_coord=lid_coord))


class CavityProjection:
    """Device-resident projection stepper.  ``step()`` returns (cg_iterations, seconds)."""

    def __init__(self, ctx, n: int, nu: float = 0.01, dt: float = None, graph: FaceGraph = None, plan=None):
        """``graph``/``plan``: this rank's local face graph and halo plan (``stormruler_amd.partition``)
        for a row-partitioned run (config 5 is 4 GPUs); ``ctx`` must then carry a communicator."""
        from . import api

        self.api, self.ctx, self.n, self.nu = api, ctx, n, nu
        h = 1.0 / n
        self.dt = dt if dt is not None else 0.2 * min(h, h * h / (6.0 * nu))
        ops = build_cavity_operators(n, graph)
        self.ops = ops
        g = ops.g
        N = g.n_cells
        self.N = N
        self.L_D = api.StencilMatrix.from_face_graph(ctx, g)
        self.L_N = api.StencilMatrix.from_face_graph(ctx, ops.g_neumann)
        mk = lambda w: api.StencilMatrix.from_face_weights(ctx, N, g.n_halo, g.inner, g.outer, *w)  # noqa: E731
        self.G = [mk(w) for w in ops.grad]
        self.D = [mk(w) for w in ops.div]
        if plan is not None and plan.n_nbrs:
            for m in [self.L_D, self.L_N, *self.G, *self.D]:
                m.set_halo(plan.nbr_rank, plan.send_ptr, plan.send_idx, plan.recv_ptr)
        self.lid = api.DeviceVector.from_numpy(ctx, ops.lid, n_halo=g.n_halo)
        vec = lambda: api.DeviceVector(ctx, N, g.n_halo)  # noqa: E731
        self.u = [vec() for _ in range(3)]
        self.us = [vec() for _ in range(3)]
        self.p, self.rhs, self.t1, self.t2 = vec(), vec(), vec(), vec()
        self.A_p = api.HipStencilOperator(self.L_N, alpha=-1.0, beta=0.0)  # -L_N p = -rhs  (SPD on the mean-free space)
        self.solver = api.CgSolver()
        # The reference's relative test is relative to the INITIAL residual |b - A x0| (Solver.hpp:135),
        # which a warm start makes tiny; a warm-started loop therefore stops on the absolute test, set
        # per step to `tol` x |rhs|.
        self.tol = 1e-8
        self.solver.relative_error_tolerance = 0.0
        self.total_time = 0.0
        self.steps = 0

    def step(self):
        api, dt, nu = self.api, self.dt, self.nu
        t0 = time.perf_counter()
        # predictor: u*_d = u_d + dt (nu L_D u_d + nu s_d - sum_e u_e .* G_e u_d)
        for d in range(3):
            self.L_D.apply(dt * nu, 1.0, self.u[d], self.us[d])
            if d == 0:
                self.us[d] += (dt * nu) * self.lid
            for e in range(3):
                self.G[e].apply(1.0, 0.0, self.u[d], self.t1)
                api.vmul_add(self.us[d], -dt, self.u[e], self.t1)
        # rhs = -(1/dt) div u*   (sign: we solve  -L_N p = -(1/dt) div u*)
        api.fill_with(self.rhs, 0.0)
        for d in range(3):
            self.D[d].apply(1.0, 0.0, self.us[d], self.t1)
            self.rhs -= (1.0 / dt) * self.t1
        self.solver.absolute_error_tolerance = self.tol * api.norm_2(self.rhs)
        ok = self.solver.solve(self.p, self.rhs, self.A_p)  # p keeps its previous value: warm start
        # corrector
        for d in range(3):
            self.G[d].apply(1.0, 0.0, self.p, self.t1)
            self.u[d] <<= self.us[d] - dt * self.t1
        self.ctx.sync()
        sec = time.perf_counter() - t0
        self.total_time += sec
        self.steps += 1
        return self.solver.iteration, sec, ok

    def divergence_norm(self) -> float:
        api = self.api
        api.fill_with(self.t2, 0.0)
        for d in range(3):
            self.D[d].apply(1.0, 0.0, self.u[d], self.t1)
            self.t2 += self.t1
        return api.norm_2(self.t2)
