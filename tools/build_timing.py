#!/usr/bin/env python3
"""Host-side cost of building the bench operator (256^3 box): python mesh, face coefficients, the library's build
(STORM_HIP_BUILD_TIMING=1 prints its stages on stderr), by thread count (STORM_HIP_BUILD_THREADS)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from stormruler_amd import api, mesh

    t = time.time()
    g = mesh.structured_box(int(sys.argv[2]))
    t_mesh = time.time() - t
    ctx = api.Context(0)
    t = time.time()
    mesh.face_coefficients(g)
    t_coef = time.time() - t
    times = []
    for _ in range(2):
        t = time.time()
        m = api.StencilMatrix.from_face_graph(ctx, g)
        times.append(time.time() - t)
        st = m.stats()
        m.close()
    print(json.dumps({"threads": os.environ.get("STORM_HIP_BUILD_THREADS", "default"), "mesh_python_s": t_mesh,
                      "face_coefficients_numpy_s": t_coef, "from_face_graph_s": times, "paired_rows": st["paired_rows"],
                      "host_cpus": os.cpu_count()}))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "256"
    for th in ("1", "4", "16"):
        env = dict(os.environ, STORM_HIP_BUILD_THREADS=th, STORM_HIP_BUILD_TIMING="1")
        subprocess.run([sys.executable, __file__, "child", n], env=env)
