#!/usr/bin/env python3
"""Full-size property of the tetrahedral operator: renumbering the cells conjugates it by a permutation -- faces keep their
order, so every row sums the same terms in the same order: y_ordered == y_file[order] BIT FOR BIT; and K CG iterations from
b = 1 leave the same residual to rounding."""
import json
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from stormruler_amd import api, host_mesh, io_tetgen  # noqa: E402

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pos, bf, cells = io_tetgen.tet_box(n3)
lab = np.ones(len(bf), np.int64)
ctx = api.Context(0)
n = 6 * n3 ** 3
x_file = np.sin(0.37 * np.arange(n))
out = {"rows": n}
ys, res = {}, {}
for mode in ("file", "morton", "hilbert"):
    hm = host_mesh.HostMesh.from_simplices(pos, bf, lab, cells)
    order = np.arange(n)
    if mode != "file":
        hm.order_cells(mode)
        order = np.ctypeslib.as_array(hm.view().global_id, shape=(n,)).copy()
    mat = hm.create_operator(ctx)
    x = api.DeviceVector.from_numpy(ctx, x_file[order])
    y = api.DeviceVector(ctx, n)
    mat.apply(-1.0, 0.0, x, y)
    yo = y.to_numpy()
    back = np.empty(n)
    back[order] = yo
    ys[mode] = back
    b = api.DeviceVector(ctx, n)
    api.fill_with(b, 1.0)
    hist = {}
    for K in (10, 50, 200):
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = K, 0.0, 0.0
        xs = api.DeviceVector(ctx, n)
        s.solve(xs, b, api.HipStencilOperator(mat, -1.0, 0.0))
        hist[K] = s.absolute_error
    res[mode] = hist
    del x, y, b, xs
    mat.close()
    hm.close()
out["spmv_bitwise_equal_to_file_order"] = {m: bool(np.array_equal(ys[m], ys["file"])) for m in ys}
out["spmv_max_rel_diff"] = {m: float(np.abs(ys[m] - ys["file"]).max() / np.abs(ys["file"]).max()) for m in ys}
out["cg_residuals"] = res
print(json.dumps(out))
