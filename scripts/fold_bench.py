#!/usr/bin/env python3
"""Tuning aid: the folded backward aggregation (gcnx_spmm_csr_pool_bwd) against the plain one, back to back."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import gcnx
from gcnx import device as D, synth
from gcnx.device import DeviceCSR, Segments
hb = synth.ecoli_batch(); f = 128
vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
ctx = gcnx.Context(0)
a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
seg = Segments(ctx, hb.graph_ptr)
rng = np.random.default_rng(0)
y = ctx.to_device(np.maximum(rng.standard_normal((hb.n, f), dtype=np.float32), 0))
dp = ctx.to_device(rng.standard_normal((len(hb.graph_ptr) - 1, f), dtype=np.float32))
out = ctx.empty((hb.n, f))
def timeit(fn, iters=200):
    for _ in range(5): fn()
    e0 = ctx.event().record()
    for _ in range(iters): fn()
    e1 = ctx.event().record()
    return e1.elapsed_ms_since(e0) / iters * 1e3
for rnd in range(3):
    print("plain %.1f us   fold %.1f us" % (timeit(lambda: D.spmm(ctx, a, y, None, out)),
                                          timeit(lambda: D.spmm_pool_bwd(ctx, a, y, seg, dp, out))), flush=True)
ctx.close()
