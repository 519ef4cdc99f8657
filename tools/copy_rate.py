"""Device copy rate of the library's own copy kernel for 128 MiB .. 2 GiB vectors."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stormruler_amd import api
ctx = api.Context(0)
for logn in (24, 25, 26, 27, 28):
    n = 1 << logn
    src, dst = api.DeviceVector(ctx, n), api.DeviceVector(ctx, n)
    api.fill_with(src, 1.0)
    for _ in range(3): dst <<= src
    ctx.timer_start()
    for _ in range(10): dst <<= src
    ms = ctx.timer_stop()/10
    print(logn, n*8/2**20, "MiB/vec", "%.1f GB/s" % (16.0*n/ms/1e6))
    del src, dst
