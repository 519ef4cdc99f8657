#!/usr/bin/env python3
"""Is `norm2` / `dot` at 2^24 rows below the copy's rate because of its fixed costs (launch ramp, ticket tail) or because a
read-only stream is slower than a read + write one?  The same kernels at 2^24 ... 2^28 rows, four reductions in flight
(begin / end: the host round trip is off the clock), beside copy and fill.  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api  # noqa: E402

ctx = api.Context(0)
out = {}
for lg in (24, 26, 28):
    n = 1 << lg
    a, b = api.DeviceVector(ctx, n), api.DeviceVector(ctx, n)
    api.fill_with(a, 1.0), api.fill_with(b, 0.5)
    reps = 40 if lg <= 26 else 12

    def timed(fn, bytes_per_call):
        for _ in range(3):
            fn()
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ms = ctx.timer_stop() / reps
        return {"us": round(ms * 1e3, 2), "TBs": round(bytes_per_call / ms / 1e9, 3)}

    pending = []

    def in_flight(vecs):  # (bench.py's form: a window of four requests)
        def go():
            pending.append(api.PendingDots(a, vecs))
            if len(pending) == 4:
                pending.pop(0).result()
        return go

    r = {}
    r["copy"] = timed(lambda: b.__ilshift__(a), 16 * n)
    r["fill"] = timed(lambda: api.fill_with(b, 0.5), 8 * n)
    r["norm2_in_flight"] = timed(in_flight([a]), 8 * n)
    while pending:
        pending.pop(0).result()
    r["dot_in_flight"] = timed(in_flight([b]), 16 * n)
    while pending:
        pending.pop(0).result()
    out[f"2^{lg}"] = r
    del a, b
print(json.dumps(out))
