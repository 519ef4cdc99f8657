#!/usr/bin/env python3
"""Stand-alone dot / norm2 at N elements, synchronous calls back to back (for a kernel trace: kernel duration vs the
host round trip between two calls)."""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256 ** 3
reps = 200
ctx = api.Context(0)
a, b = api.DeviceVector(ctx, N), api.DeviceVector(ctx, N)
api.fill_with(a, 1.0)
api.fill_with(b, 1.001)
for name, fn, bpe in (("dot", lambda: api.dot_product(a, b), 16), ("norm2", lambda: api.norm_2(a), 8)):
    for _ in range(10):
        fn()
    ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t) / reps
    print(f"{name}: {dt * 1e6:.1f} us per call, {bpe * N / dt / 1e9:.0f} GB/s, frac {bpe * N / dt / 8e12:.3f}")
