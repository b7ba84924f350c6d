import sys, os
sys.path.insert(0, "gcn-string_amd")
import numpy as np, gcnx
ctx = gcnx.Context(0)
for mb in (10.7, 21.4, 100):
    n = int(mb * 1e6 / 4)
    a = ctx.to_device(np.ones(n, np.float32)); b = ctx.empty(n)
    for _ in range(3): ctx._ck(ctx.lib.gcnx_d2d(ctx.h, b.ptr, a.ptr, n * 4))
    e0 = ctx.event().record()
    for _ in range(200): ctx._ck(ctx.lib.gcnx_d2d(ctx.h, b.ptr, a.ptr, n * 4))
    e1 = ctx.event().record()
    us = e1.elapsed_ms_since(e0) / 200 * 1e3
    print(f"d2d copy {mb} MB: {us:.1f} us  {2*n*4/us/1e3:.0f} GB/s (read+write)")
    # sgd-like streaming kernel: p -= lr*g  (2 reads + 1 write)
    e0 = ctx.event().record()
    for _ in range(200): ctx._ck(ctx.lib.gcnx_sgd(ctx.h, b.ptr, a.ptr, n, 0.0))
    e1 = ctx.event().record()
    us = e1.elapsed_ms_since(e0) / 200 * 1e3
    print(f"sgd kernel {mb} MB: {us:.1f} us  {3*n*4/us/1e3:.0f} GB/s")
ctx.close()
