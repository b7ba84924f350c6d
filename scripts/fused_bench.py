#!/usr/bin/env python3
"""Fused GCNConv launch times at config-2 shapes (tuning aid): forward, backward, the two-gradient GEMM; HIP events.
GCNX_FUSED_DBG (tuning build only) ablates phases: 1 no gather, 2 no MFMA, 4 no weight load."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import gcnx
from gcnx import device as D, synth
from gcnx.device import DeviceCSR, Segments

hb = synth.ecoli_shard(0, 32, 128, seed=1); f = 128
vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
ctx = gcnx.Context(0)
a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
rng = np.random.default_rng(0)
x = ctx.to_device(hb.x); w = ctx.to_device((rng.standard_normal((f, f)) / 11).astype(np.float32)); b = ctx.zeros(f)
out = ctx.empty((hb.n, f)); s = ctx.empty((hb.n, f)); wt = ctx.empty((f, f)); dz2 = ctx.empty((hb.n, f)); dz1 = ctx.empty((hb.n, f))
seg = Segments(ctx, hb.graph_ptr); dp = ctx.to_device(rng.standard_normal((32, f), dtype=np.float32))
g = ctx.zeros(2 * f * f + f); scratch = ctx.empty(D.gcn_conv_bwd_scratch_floats(ctx, hb.n, f))
at = a.transpose()

def timeit(fn, iters=50):
    for _ in range(5): fn()
    e0 = ctx.event().record()
    for _ in range(iters): fn()
    e1 = ctx.event().record()
    return e1.elapsed_ms_since(e0) / iters * 1e3

print("fwd  (S, W^T out): %.1f us" % timeit(lambda: D.gcn_conv_fwd(ctx, a, x, w, b, out, act="relu", s=s, wt=wt)))
print("fwd  (inference) : %.1f us" % timeit(lambda: D.gcn_conv_fwd(ctx, a, x, w, b, out, act="relu")))
print("bwd              : %.1f us" % timeit(lambda: D.gcn_conv_bwd_pool(ctx, at, out, seg, dp, w, s, dz2, dz1, db1=g.flat(2 * f * f, f), scratch=scratch, w2t=wt)))
tp, tc = ctx.zeros((D.pool_tile_rows(hb.n, 32), f)), ctx.zeros((D.pool_tile_rows(hb.n, 32), f))
w3 = ctx.to_device((rng.standard_normal((f, 2)) / 1e3).astype(np.float32)); b3 = ctx.zeros(2)
yl = ctx.to_device(np.eye(2, dtype=np.float32)[rng.integers(0, 2, 32)])
print("fwd  (S, W^T, pool partials): %.1f us" % timeit(lambda: D.gcn_conv_fwd(ctx, a, x, w, b, out, act="relu", s=s, wt=wt, pool=(seg, tp, tc))))
ha = D.head_args(seg, tp, tc, ctx.empty((32, f)), ctx.empty((32, f)), w3, b3, yl, 32.0, ctx.empty((32, 2)), ctx.zeros(2), ctx.empty((f, 2)),
                 ctx.empty(2), ctx.empty(f), ctx.empty((32, f)), ctx.empty((32, f)))
print("bwd  (head inside): %.1f us" % timeit(lambda: D.gcn_conv_bwd_pool(ctx, at, out, seg, None, w, s, dz2, dz1, db1=g.flat(2 * f * f, f), scratch=scratch, w2t=wt, head=ha)))
print("dw2              : %.1f us" % timeit(lambda: D.gemm_dw2(ctx, s, dz1, g.flat(0, f * f, (f, f)), s, dz2, g.flat(f * f, f * f, (f, f)), grads=g)))
h = ctx.empty((hb.n, f))
print("gemm + spmm      : %.1f us" % timeit(lambda: (D.gemm(ctx, x, w, None, h), D.spmm(ctx, a, h, b, out, act="relu"))))
ctx.close()
