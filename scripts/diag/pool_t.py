#!/usr/bin/env python3
"""Config-3 timing: gcnx_spmm_csr_relu_bits + pool partials against gcnx_spmm_csr_relu_bits_pool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import gcnx
from gcnx import device as D, synth
from gcnx.device import DeviceCSR, Segments
sizes, pairs = synth.block_diag_plan()
hb = synth.block_diag_shard(0, len(sizes), sizes, pairs, 256, seed=2, with_x=False)
vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
ctx = gcnx.Context(0)
a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
seg = Segments(ctx, hb.graph_ptr)
n, f, b = hb.n, 256, seg.n_graphs
h = ctx.to_device(np.random.default_rng(0).standard_normal((n, f), dtype=np.float32)); bias = ctx.zeros(f)
y = ctx.empty((n, f)); bits = ctx.zeros((f // 32) * n, np.int32); pooled = ctx.zeros((b, f)); cnt = ctx.zeros((b, f))
w3 = ctx.to_device(np.random.default_rng(1).standard_normal((f, 2)).astype(np.float32)); b3 = ctx.zeros(2)
yl = np.zeros((b, 2), np.float32); yl[:, 0] = 1
dy = ctx.to_device(yl); probs = ctx.empty((b, 2)); la = ctx.zeros(2); dw = ctx.empty((f, 2)); db = ctx.empty(2); dpo = ctx.empty((b, f)); dbr = ctx.empty(f)
def two():
    D.spmm_relu_bits(ctx, a, h, bias, y, bits)
    D.pool_dense_softmax_cce(ctx, seg, y, pooled, w3, b3, dy, probs, la, float(b), dw=dw, db=db, dpooled=dpo, db_relu=dbr, cce="logits", mode="sum", argmax=None)
def one():
    D.spmm_relu_bits_pool(ctx, a, h, bias, y, bits, seg, pooled, cnt, "sum")
for name, fn in (("relu_bits + pool + head", two), ("relu_bits_pool (no head)", one), ("relu_bits + pool + head", two), ("relu_bits_pool (no head)", one)):
    for _ in range(3): fn()
    e0 = ctx.event().record()
    for _ in range(20): fn()
    print(f"{name:28s} {ctx.event().record().elapsed_ms_since(e0) / 20 * 1e3:8.1f} us", flush=True)
ctx.close()
