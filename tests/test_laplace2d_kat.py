"""The reference's benchmark-as-test (tests/benchmark/BitternLaplace2D.cpp:359-424): 1000 sweeps of a 9-point
Jacobi iteration on an N x N grid, known answers for N = 100, 150, 200.  Not the hot path itself (dense Bittern
matrices in the reference) but three reference-held numbers that an SpMV + `norm_2` must reproduce: the sweep is one
sparse operator (interior rows: the 9-point average; boundary rows: identity), the checked value is
|u_new - u_old|_2.  Rows have 8 neighbours, so the operator takes the general fp64-record kernel at width 8."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle


def _sweep_matrix(n):
    idx = np.arange(n * n).reshape(n, n)
    rows, cols, vals = [], [], []
    inner = idx[1:-1, 1:-1].ravel()
    for di, dj, w in ((-1, 0, 4.0), (1, 0, 4.0), (0, -1, 4.0), (0, 1, 4.0), (-1, -1, 1.0), (-1, 1, 1.0), (1, -1, 1.0), (1, 1, 1.0)):
        rows.append(inner)
        cols.append(idx[1 + di:n - 1 + di, 1 + dj:n - 1 + dj].ravel())
        vals.append(np.full(inner.size, w / 20.0))
    border = np.setdiff1d(idx.ravel(), inner)
    rows.append(border), cols.append(border), vals.append(np.ones(border.size))
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n * n, n * n)).tocsr()


def _u0(n):
    x = np.linspace(0.0, np.pi, n)
    u = np.zeros((n, n))
    u[:, 0] = np.sin(x)
    u[:, n - 1] = np.sin(x) * np.exp(-np.pi)
    return u.ravel()


@pytest.mark.parametrize("case", [0, 1, 2])
def test_numpy_restatement_and_oracle_operator(golden, case):
    c = golden["laplace2d_benchmark"]["cases"][case]
    n, its, tol = c["N"], golden["laplace2d_benchmark"]["num_iterations"], golden["laplace2d_benchmark"]["tolerance"]
    u = _u0(n).reshape(n, n)
    for _ in range(its):  # the sweep as the reference's NumPy variant spells it (:318-326)
        u_old = u.copy()
        u[1:-1, 1:-1] = (4.0 * (u_old[0:-2, 1:-1] + u_old[2:, 1:-1] + u_old[1:-1, 0:-2] + u_old[1:-1, 2:]) +
                         1.0 * (u_old[0:-2, 0:-2] + u_old[0:-2, 2:] + u_old[2:, 0:-2] + u_old[2:, 2:])) / 20.0
    err = np.sqrt(np.sum((u - u_old) ** 2))
    assert abs(err - c["expected_error"]) <= tol * c["expected_error"]
    if case == 0:  # the same sweep as a sparse operator through the oracle's CSR apply + norm2
        op = oracle.CsrOperator(_sweep_matrix(n))
        v = _u0(n)
        for _ in range(its):
            w = op.apply(v)
            e2 = oracle.norm2(w - v)
            v = w
        assert abs(e2 - err) <= 1e-12 * err


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1, 2])
def test_hip_path_reproduces_the_reference_known_answers(golden, case):
    from stormruler_amd import api

    c = golden["laplace2d_benchmark"]["cases"][case]
    n, its, tol = c["N"], golden["laplace2d_benchmark"]["num_iterations"], golden["laplace2d_benchmark"]["tolerance"]
    ctx = api.Context(0)
    mat = api.StencilMatrix.from_csr(ctx, _sweep_matrix(n))
    st = mat.stats()
    assert st["max_row_len"] == 8 and st["value_dictionary_size"] == 0  # general kernel, width 8
    u, w, d = api.DeviceVector.from_numpy(ctx, _u0(n)), api.DeviceVector(ctx, n * n), api.DeviceVector(ctx, n * n)
    for _ in range(its):
        mat.apply(1.0, 0.0, u, w)          # u_new = M u
        d <<= w - u
        u, w = w, u
    err = api.norm_2(d)
    assert abs(err - c["expected_error"]) <= tol * c["expected_error"]
    ctx.close()
