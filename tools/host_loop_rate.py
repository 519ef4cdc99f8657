#!/usr/bin/env python3
"""What a HOST loop costs: the reference's CG body typed statement by statement against Storm.hpp by a user
(tests/cpp/poisson_driver.cpp `user-cg`) at n^3, with the library's lazy statements (csrc/lazy.hip: `x += alpha p;
r -= alpha z; <r, r>` one kernel, `x += alpha p; p <<= r + beta p; z = A p; <p, z>` the fused CG step) and without (every statement a launch
when it is called), against the library's own device-resident CG loop.  us per iteration, fixed iteration count."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "poisson_driver")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200


def run(kind, mode):
    p = subprocess.run([DRIVER, str(n), kind, mode], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, DRIVER_FIXED_ITERATIONS=str(iters)))
    assert p.returncode == 0, p.stdout + p.stderr
    for ln in p.stdout.splitlines():
        if "timed_solve_seconds" in ln:
            d = json.loads(ln)
            return d["timed_solve_seconds"] / d["timed_iterations"] * 1e6
    raise RuntimeError(p.stdout)


out = {"n": n, "iterations": iters,
       "device_loop_us_per_iteration": run("cg", "native"),
       "host_loop_lazy_statements_us_per_iteration": run("user-cg", "native"),
       "host_loop_eager_statements_us_per_iteration": run("user-cg", "eager")}
out["host_loop_lazy_over_device_loop"] = out["host_loop_lazy_statements_us_per_iteration"] / out["device_loop_us_per_iteration"]
out["host_loop_eager_over_device_loop"] = out["host_loop_eager_statements_us_per_iteration"] / out["device_loop_us_per_iteration"]
print(json.dumps(out))
