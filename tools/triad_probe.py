import sys; sys.path.insert(0, ".")
from diaglib_amd import capi
ctx = capi.Context()
for ln in (1 << 26, 1 << 28):
    print(ln, round(ctx.stream_triad(ln, 5), 1), "GB/s")
