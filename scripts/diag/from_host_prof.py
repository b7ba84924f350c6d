#!/usr/bin/env python3
"""Where DeviceBatch.from_host spends its time (host loader path, E. coli-shaped batch of 32)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np, scipy.sparse as sp
import gcnx
from gcnx import synth, DisjointLoader, Graph, ListDataset, device as D
rng = np.random.default_rng(0)
graphs = []
for _ in range(32):
    n, u, v = synth.ecoli_graph_pairs(rng)
    a = sp.coo_matrix((np.ones(u.size), (u, v)), shape=(n, n)).tocsr()
    a = ((a + a.T + sp.identity(n)) > 0).astype(np.float32).tocsr()
    y = np.zeros(2, np.float32); y[0] = 1
    graphs.append(Graph(x=rng.standard_normal((n, 128), dtype=np.float32), a=a, y=y))
ctx = gcnx.Context(0)
loader = DisjointLoader(ListDataset(graphs), batch_size=32, epochs=None, shuffle=False)
(x, a, i), y = next(loader)
n = x.shape[0]
T = {}
def tick(name, t0):
    ctx.sync(); T.setdefault(name, []).append(time.perf_counter() - t0)
for rep in range(8):
    t = time.perf_counter(); seg = D.Segments.from_ids(ctx, i); tick("segments", t)
    t = time.perf_counter(); dx = ctx.to_device(x, np.float32); tick("x h2d (%.1f MB)" % (x.nbytes / 1e6), t)
    t = time.perf_counter(); r = np.ascontiguousarray(a.indices[:, 0]); c = np.ascontiguousarray(a.indices[:, 1]); tick("host split of the pairs", t)
    t = time.perf_counter(); csr = D.DeviceCSR.from_coo(ctx, a.indices, a.values, n, graph_ptr=seg.host); tick("from_coo", t)
    t = time.perf_counter(); csr2 = csr.gcn_norm("spektral"); tick("gcn_norm", t)
    t = time.perf_counter(); _ = csr2.symmetric; tick("symmetric (inspect)", t)
    t = time.perf_counter(); dy = ctx.to_device(y, np.float32); tick("y h2d", t)
    t = time.perf_counter(); del dx, csr, csr2, dy, seg; tick("free", t)
for k, v in T.items():
    print(f"{k:32s} {1e3 * np.median(v[2:]):7.3f} ms")
