import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
t=time.time(); from stormruler_amd import api, mesh; print("import", time.time()-t)
t=time.time(); g = mesh.structured_box(256); print("mesh", time.time()-t)
ctx = api.Context(0)
t=time.time(); coef, b_coef = api.face_coefficients(g) if hasattr(api,"face_coefficients") else (None,None); print("coef", time.time()-t)
for d in (0, 2):
    ctx.set_option("spmv_dict", d)
    t=time.time(); mat = api.StencilMatrix.from_face_graph(ctx, g); ctx.sync(); print("from_face_graph dict=%d"%d, time.time()-t)
    mat.close()
print("cpus", os.cpu_count())
