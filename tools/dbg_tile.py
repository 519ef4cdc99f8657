import sys, numpy as np
sys.path.insert(0, '.')
from stormruler_amd import api, mesh
ctx = api.Context(0)
for shape in [(20,6,9),(256,8,8),(128,16,11),(64,40,9),(100,22,13),(34,34,17),(256,12,10)]:
    g = mesh.structured_box(*shape)
    x = np.sin(0.37*np.arange(g.n_cells)) + 1e-3*np.cos(1.7*np.arange(g.n_cells))
    def run(tile):
        ctx.set_option("spmv_canon_tile", tile); ctx.set_option("spmv_canon_tile_min_rows", 0)
        m = api.StencilMatrix.from_face_graph(ctx, g)
        xv, yv = api.DeviceVector.from_numpy(ctx, x), api.DeviceVector(ctx, x.size)
        m.apply(-0.7, 0.3, xv, yv)
        y = yv.to_numpy(); st = m.stats(); m.close(); return y, st
    y0, st0 = run(0)
    for tz in (4, 2):
        y1, st = run(tz)
        bad = np.flatnonzero(y0 != y1)
        nx, ny, nz = shape
        print(shape, tz, "tiled", st["tiled_planes"], "blocks", st["spmv_blocks"], "bad", bad.size, "of", y0.size)
        if bad.size:
            print("  first bad rows", bad[:12], "q", (bad[:12] % (nx*ny)), "plane", bad[:12] // (nx*ny))
            print("  rel diff max", np.abs(y0[bad]-y1[bad]).max()/np.abs(y0).max())
            pl = np.unique(bad // (nx*ny)); print("  planes", pl[:20]); qq = np.unique(bad % (nx*ny)); print("  q range", qq.min(), qq.max(), qq.size)
