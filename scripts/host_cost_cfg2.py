#!/usr/bin/env python3
"""Host cost of one eager config-2 train step (GCN2(use_graph="auto")): time to ENQUEUE a step against the time the GPU needs for it.
    python scripts/host_cost_cfg2.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import gcnx
from gcnx import synth
from gcnx.device import DeviceCSR, Segments
from gcnx.models import DeviceBatch, GCN2
ctx = gcnx.Context(0)
hb = synth.ecoli_shard(0, 32, 128, seed=1)
vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
for opt in ("auto", True):
    m = GCN2(ctx, 2, hidden=128, seed=0, use_graph=opt)
    for _ in range(50): m.train_step(batch, None, lr=0.02, fetch=False)
    ctx.sync()
    # host cost: enqueue a short burst into an EMPTY queue (the GPU cannot be the limit for the first steps of a burst)
    costs = []
    for _ in range(20):
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(8): m.train_step(batch, None, lr=0.02, fetch=False)
        costs.append((time.perf_counter() - t0) / 8)
        ctx.sync()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(2000): m.train_step(batch, None, lr=0.02, fetch=False)
    ctx.sync(); wall = (time.perf_counter() - t0) / 2000
    print(f"use_graph={opt}: host enqueue {np.median(costs) * 1e6:.1f} us per step (median of 20 bursts of 8), steady state {wall * 1e6:.1f} us per step")
