"""Generates the committed golden fixtures tests/golden/*.npz.

PARITY UNPINNED: the reference (Sum02dean/GCN-STRING) ships no tests, fixtures or golden
vectors for this path and its Spektral/TensorFlow dependencies are not importable here, so the
vectors come from the build's own fp64 numpy restatement (oracle/gcn_oracle.py, following
SURVEY.md 8.A).  A vector is written only if two independent implementations that ARE
importable in the build container agree with it to <= 1e-10: scipy.sparse (SpMM,
normalisation) and torch-CPU fp64 autograd (loss and every gradient).  torch/scipy are
generation-time checks only; the fixtures are plain arrays (inputs fp32-representable so the
GPU sees bit-identical inputs; expected outputs rounded to fp32).

Run from the repo root:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))

from oracle import gcn_oracle as O  # noqa: E402
from gcnx import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def torch_gcn2(params, x, rowptr, colidx, vals, gp, y, pool, cce_mode):
    import torch

    n = x.shape[0]
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    v = np.ones(len(colidx)) if vals is None else vals
    a = torch.sparse_coo_tensor(np.stack([rows, colidx]), torch.tensor(v, dtype=torch.float64), (n, n)).coalesce()
    tp = {k: torch.tensor(p, dtype=torch.float64, requires_grad=True) for k, p in params.items()}
    xt = torch.tensor(x, dtype=torch.float64)
    y1 = torch.relu(torch.sparse.mm(a, xt @ tp["w1"]) + tp["b1"])
    y2 = torch.relu(torch.sparse.mm(a, y1 @ tp["w2"]) + tp["b2"])
    pooled = []
    for g in range(len(gp) - 1):
        seg = y2[gp[g]:gp[g + 1]]
        pooled.append(seg.sum(0) if pool == "sum" else seg.mean(0) if pool == "avg" else seg.max(0).values)
    pooled = torch.stack(pooled)
    logits = pooled @ tp["w3"] + tp["b3"]
    probs = torch.softmax(logits, dim=1)
    yt = torch.tensor(y, dtype=torch.float64)
    if cce_mode == "logits":      # softmax_cross_entropy_with_logits (what Keras runs inside tf.function)
        loss = -(yt * torch.log_softmax(logits, dim=1)).sum(1).mean()
    else:                         # eager Keras: renormalise, clip, log
        pc = torch.clamp(probs / probs.sum(1, keepdim=True), 1e-7, 1 - 1e-7)
        loss = -(yt * torch.log(pc)).sum(1).mean()
    loss.backward()
    return loss.item(), probs.detach().numpy(), {k: t.grad.numpy() for k, t in tp.items()}, y2.detach().numpy()


def make_case(name, hb, hidden, weighted, pool="sum", seed=0, w3_scale=1.0):
    """hb: synth.HostBatch (x fp32).  Writes golden/<name>.npz."""
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    rowptr, colidx = hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64)
    vals32 = synth.gcn_norm_host(hb.rowptr, hb.colidx) if weighted else None
    vals = None if vals32 is None else vals32.astype(np.float64)
    if weighted:  # fp64 normalisation agrees with the oracle's own gcn_filter and with scipy
        v_or = O.gcn_filter_csr(rowptr, colidx, None, "spektral")
        a01 = sp.csr_matrix((np.ones(len(colidx)), colidx, rowptr), shape=(hb.n, hb.n))
        a01.setdiag(0); a01.eliminate_zeros()          # structural form adds I itself
        ref = O.gcn_filter_scipy(a01 + sp.identity(hb.n), "spektral")
        assert rel(v_or, ref.data) < 1e-12, "gcn_filter: oracle vs scipy"
        assert rel(vals32.astype(np.float64), v_or) < 1e-6
    x = hb.x.astype(np.float64)
    y = hb.y.astype(np.float64)
    params = O.gcn2_init(rng, hb.f, hidden, 2)
    params = {k: v.astype(np.float32).astype(np.float64) for k, v in params.items()}
    params["b1"] = (0.1 * rng.standard_normal(hidden)).astype(np.float32).astype(np.float64)
    params["b2"] = (0.1 * rng.standard_normal(hidden)).astype(np.float32).astype(np.float64)
    params["b3"] = (0.1 * rng.standard_normal(2)).astype(np.float32).astype(np.float64)
    params["w3"] = (params["w3"] * w3_scale).astype(np.float32).astype(np.float64)
    csr = (rowptr, colidx, vals)
    # "logits": the loss train_step computes under tf.function (keys loss, g_*); "probs": the eager clip form
    # (keys loss_probs, gp_*)
    loss, acc, grads, cache = O.gcn2_loss_and_grads(params, x, csr, hb.graph_ptr, y, pool, cce_mode="logits")
    loss_p, _, grads_p, _ = O.gcn2_loss_and_grads(params, x, csr, hb.graph_ptr, y, pool, cce_mode="probs")
    # independent check 1: scipy SpMM
    a = sp.csr_matrix((np.ones(len(colidx)) if vals is None else vals, colidx, rowptr), shape=(hb.n, hb.n))
    h1 = x @ params["w1"]
    assert rel(O.spmm_csr(rowptr, colidx, vals, h1), a @ h1) < 1e-12, "spmm: oracle vs scipy"
    # independent check 2: torch fp64 autograd
    for mode, l_or, g_or in (("logits", loss, grads), ("probs", loss_p, grads_p)):
        t_loss, t_probs, t_grads, t_y2 = torch_gcn2(params, x, rowptr, colidx, vals, hb.graph_ptr, y, pool, mode)
        assert abs(l_or - t_loss) < 1e-10 * max(1, abs(t_loss)), (mode, l_or, t_loss)
        assert rel(cache["probs"], t_probs) < 1e-10
        assert rel(cache["y2"], t_y2) < 1e-10
        for k in g_or:
            if np.max(np.abs(t_grads[k])) == 0 and np.max(np.abs(g_or[k])) == 0:
                continue
            assert rel(g_or[k], t_grads[k]) < 1e-10, (mode, k, rel(g_or[k], t_grads[k]))
    out = {
        "x": hb.x.astype(np.float32), "rowptr": hb.rowptr.astype(np.int32), "colidx": hb.colidx.astype(np.int32),
        "graph_ptr": hb.graph_ptr.astype(np.int32), "y": hb.y.astype(np.float32),
        "weighted": np.array(int(weighted)), "pool": np.array(pool), "lr": np.array(0.02),
        "y2": cache["y2"].astype(np.float32), "pooled": cache["pooled"].astype(np.float32),
        "probs": cache["probs"], "logits": cache["logits"], "loss": np.array(loss), "loss_probs": np.array(loss_p),
        "acc": np.array(acc),
    }
    if vals32 is not None:
        out["vals"] = vals32
    for k in O.GCN2_PARAM_ORDER:
        out["p_" + k] = params[k].astype(np.float32)
        out["g_" + k] = grads[k].astype(np.float32)
        out["gp_" + k] = grads_p[k].astype(np.float32)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N={hb.n} nnz={hb.nnz} B={hb.n_graphs} F={hb.f} H={hidden} loss={loss:.6f} (probs form {loss_p:.6f}) "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


def tiny_hostbatch(seed=0, f=32, n_graphs=16):
    """config 1 through the DisjointLoader-equivalent collate of the oracle."""
    graphs = synth.tiny_graphs(n_graphs, f, seed)
    x, (idx, val, shape), i, y = O.disjoint_collate(graphs)
    rowptr, colidx = O.coo_to_csr(idx, shape[0])
    gp = O.graph_ptr_from_ids(i, len(graphs))
    return synth.HostBatch(x.astype(np.float32), rowptr.astype(np.int32), colidx.astype(np.int32), None,
                           gp.astype(np.int32), y.astype(np.float32))


def edge_hostbatch(seed=7, f=10):
    """Edge cases: a single-node graph, a graph whose middle node has NO entries at all (empty
    CSR row -- only reachable when self-loops are absent), a feature width that is not a
    multiple of 4, a two-node graph."""
    rng = np.random.default_rng(seed)
    # graph A: 1 node, self-loop.  graph B: path 0-1-2 with self loops on 0,2 only and node 3
    # isolated with no self-loop (empty row).  graph C: 2 nodes fully connected + loops.
    rows = [0, 1, 1, 2, 2, 3, 3, 5, 5, 6, 6]
    cols = [0, 1, 2, 1, 3, 2, 3, 5, 6, 5, 6]
    n = 7
    key = np.unique(np.array(rows) * n + np.array(cols))
    rows, cols = key // n, key % n
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    gp = np.array([0, 1, 5, 7], dtype=np.int32)
    y = np.array([[1, 0], [0, 1], [1, 0]], dtype=np.float32)
    x = rng.standard_normal((n, f)).astype(np.float32)
    return synth.HostBatch(x, rowptr, cols.astype(np.int32), None, gp, y)


def main():
    make_case("gcn2_cfg1_tiny_weighted", tiny_hostbatch(0, 32), 32, True)
    make_case("gcn2_cfg1_tiny_unweighted_max", tiny_hostbatch(1, 32, 8), 32, False, pool="max")
    make_case("gcn2_ecoli_mini_f16", synth.ecoli_batch(3, 16, seed=11), 16, True)
    make_case("gcn2_tiny_f128_avg", tiny_hostbatch(2, 128, 6), 128, True, pool="avg")
    make_case("gcn2_edge_cases_f10", edge_hostbatch(), 6, False)
    pl = synth.power_law_batch(n_graphs=2, graph_size=512, f=32, seed=3, max_deg=256)
    pl.x *= 0.02
    make_case("gcn2_powerlaw_mini", pl, 32, True)
    # saturated logits (|z_0 - z_1| up to several hundred): the clip of the eager form is active, the two Keras code
    # paths give different losses and gradients (the clipped graphs get NO gradient in the "probs" form)
    make_case("gcn2_saturated_logits", tiny_hostbatch(5, 16, 8), 16, True, seed=4, w3_scale=2.0)    # |z0 - z1| 40..70
    make_case("gcn2_saturated_mixed", tiny_hostbatch(5, 16, 8), 16, True, seed=4, w3_scale=0.55)   # 6 of 8 graphs clipped


if __name__ == "__main__":
    main()
