import sys, os, json
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api
ctx = api.Context(0)
N = 256**3
def build(offs):
    diags = [np.full(N - abs(o), 1.0 + 0.0) for o in offs]
    a = sp.diags(diags, offs, shape=(N, N), format="csr")
    return api.StencilMatrix.from_csr(ctx, a)
x = api.DeviceVector.from_numpy(ctx, np.sin(0.37*np.arange(N))); y = api.DeviceVector(ctx, N)
out = {}
for name, offs in (("near", [-3,-2,-1,1,2,3]), ("stencil", [-65536,-256,-1,1,256,65536]), ("mid", [-768,-512,-256,256,512,768]), ("far", [-3*65536,-2*65536,-65536,65536,2*65536,3*65536])):
    m = build(offs)
    st = m.stats()
    for _ in range(5): m.apply(-1.0, 0.0, x, y)
    ts = []
    for r in range(3):
        ctx.timer_start()
        for _ in range(40): m.apply(-1.0, 0.0, x, y)
        ts.append(ctx.timer_stop()/40)
    out[name] = {"ms": min(ts), "dict": st["value_dictionary_size"], "record_bytes": st["record_bytes"], "GBs": (st["record_bytes"]+16*N)/min(ts)/1e6}
    m.close()
print(json.dumps(out))
