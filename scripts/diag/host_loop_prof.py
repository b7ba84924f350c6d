#!/usr/bin/env python3
"""Per-phase host time of the reference-style loop: collate | from_host | train_step (eager: every batch a new shape)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np, scipy.sparse as sp
import gcnx
from gcnx import synth, DisjointLoader, Graph, ListDataset
from gcnx.models import DeviceBatch, GCN2
rng = np.random.default_rng(0)
graphs = []
for _ in range(256):
    n, u, v = synth.ecoli_graph_pairs(rng)
    a = sp.coo_matrix((np.ones(u.size), (u, v)), shape=(n, n)).tocsr()
    a = ((a + a.T + sp.identity(n)) > 0).astype(np.float32).tocsr()
    y = np.zeros(2, np.float32); y[int(rng.integers(0, 2))] = 1
    graphs.append(Graph(x=rng.standard_normal((n, 128), dtype=np.float32), a=a, y=y))
ctx = gcnx.Context(0)
model = GCN2(ctx, 2, hidden=128, use_graph=False, seed=0)
loader = DisjointLoader(ListDataset(graphs), batch_size=32, epochs=4, shuffle=True, seed=1)
T = {"collate": [], "from_host": [], "train_step (queue)": [], "sync": []}
it = iter(loader)
k = 0
while True:
    t0 = time.perf_counter()
    try: inputs, target = next(it)
    except StopIteration: break
    t1 = time.perf_counter()
    batch = DeviceBatch.from_host(ctx, inputs, target, normalize="spektral")
    t2 = time.perf_counter()
    model.train_step(batch, None, lr=0.01, fetch=False)
    t3 = time.perf_counter()
    ctx.sync()
    t4 = time.perf_counter()
    if k >= 8:
        T["collate"].append(t1 - t0); T["from_host"].append(t2 - t1); T["train_step (queue)"].append(t3 - t2); T["sync"].append(t4 - t3)
    k += 1
for name, v in T.items():
    print(f"{name:22s} median {1e3*np.median(v):7.3f} ms   mean {1e3*np.mean(v):7.3f} ms")
