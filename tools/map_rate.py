#!/usr/bin/env python3
"""Rate of the element map `out <<= map(func, mats...)` (storm_hip_map, csrc/blas1.hip map_kernel) at 2^24 rows against a
plain copy: python tools/map_rate.py"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from stormruler_amd import api
ctx = api.Context(0)
N = 1 << 24
a, b, c = (api.DeviceVector(ctx, N) for _ in range(3))
api.fill_with(b, 0.3); api.fill_with(c, 0.7)
progs = {"dF_dc (13 ops)": api.map(lambda x: 2.0 * x * (x - 1.0) * (2.0 * x - 1.0), b),
         "x + 1 (3 ops)": api.map(lambda x: x + 1.0, b),
         "two inputs (7 ops)": api.map(lambda x, y: x * y + 0.5 * (x - y), b, c),
         "copy": None}
for name, m in progs.items():
    for _ in range(3):
        if m is None: a <<= b
        else: a <<= m
    ctx.timer_start()
    for _ in range(20):
        if m is None: a <<= b
        else: a <<= m
    ms = ctx.timer_stop() / 20
    streams = 2 if (m is None or len(m.xs) == 1) else 3
    print(name, round(ms * 1e3, 1), "us", round(8 * streams * N / (ms * 1e-3) / 1e9 / 8000, 3), "of peak")
