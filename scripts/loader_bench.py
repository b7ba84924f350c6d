#!/usr/bin/env python3
"""Epoch throughput WITH batch assembly (the resident-batch number of bench.py leaves it out): E. coli-shaped
graphs (config 2 sizes), batch 32, the 2-layer GCN train step.
   host   : DisjointLoader (host vstack/block_diag/find) + DeviceBatch.from_host (H2D, COO->CSR, gcn_filter) per batch
   device : DeviceDataset (uploaded and filtered once) + DeviceDisjointLoader (one gcnx_collate launch per batch)
    python scripts/loader_bench.py [--graphs 256] [--epochs 2]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import scipy.sparse as sp
import gcnx
from gcnx import DisjointLoader, Graph, ListDataset, synth, DeviceDataset, DeviceDisjointLoader
from gcnx.models import DeviceBatch, GCN2

ap = argparse.ArgumentParser()
ap.add_argument("--graphs", type=int, default=256)
ap.add_argument("--epochs", type=int, default=2)
ap.add_argument("--f", type=int, default=128)
args = ap.parse_args()
rng = np.random.default_rng(0)
graphs = []
for _ in range(args.graphs):
    n, u, v = synth.ecoli_graph_pairs(rng)
    a = sp.coo_matrix((np.ones(u.size), (u, v)), shape=(n, n)).tocsr()
    a = ((a + a.T + sp.identity(n)) > 0).astype(np.float32).tocsr()
    y = np.zeros(2, np.float32); y[int(rng.integers(0, 2))] = 1
    graphs.append(Graph(x=rng.standard_normal((n, args.f), dtype=np.float32), a=a, y=y))
ds = ListDataset(graphs)
ctx = gcnx.Context(0)
for mode in ("host", "device"):
    model = GCN2(ctx, 2, hidden=args.f, use_graph=False, seed=0)
    t_setup = time.perf_counter()
    if mode == "host":
        loader = DisjointLoader(ds, batch_size=32, epochs=args.epochs + 1, shuffle=True, seed=1)
    else:
        loader = DeviceDisjointLoader(DeviceDataset(ctx, ds, normalize="spektral"), batch_size=32, epochs=args.epochs + 1,
                                      shuffle=True, seed=1)
    ctx.sync(); t_setup = time.perf_counter() - t_setup
    spe = loader.steps_per_epoch
    seen, t0 = 0, None
    for step, (inputs, target) in enumerate(loader):
        if step == spe:                                   # first epoch = warm-up
            ctx.sync(); t0 = time.perf_counter(); seen = 0
        batch = DeviceBatch.from_host(ctx, inputs, target, normalize="spektral") if mode == "host" else inputs
        model.train_step(batch, None, lr=0.01, fetch=False)
        seen += batch.n_graphs
    ctx.sync()
    dt = time.perf_counter() - t0
    print(f"{mode:6s}: setup {t_setup*1e3:7.1f} ms   {seen} graphs in {dt*1e3:7.1f} ms -> {seen/dt:9.0f} graphs/s "
          f"({dt/(spe*args.epochs)*1e3:.3f} ms per batch incl. assembly)", flush=True)
ctx.close()
