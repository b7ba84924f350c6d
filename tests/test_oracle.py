"""CPU tests of the oracle itself (PARITY UNPINNED -- see oracle/gcn_oracle.py): the numpy
restatement against scipy / torch-CPU autograd / finite differences, the committed goldens,
and the C restatement against the numpy one."""
import numpy as np
import pytest

from conftest import GOLDEN, golden_batch, load_golden, rel_err
from oracle import gcn_oracle as O


def _rand_csr(rng, n, p=0.1, self_loops=True, symmetric=True):
    a = rng.random((n, n)) < p
    if symmetric:
        a = np.triu(a, 1); a = a | a.T
    if self_loops:
        a = a | np.eye(n, dtype=bool)
    else:
        a = a & ~np.eye(n, dtype=bool)
    rows, cols = np.nonzero(a)
    rowptr = np.zeros(n + 1, np.int64); np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
    return rowptr, cols.astype(np.int64)


def test_spmm_matches_scipy_weighted_and_unweighted():
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    rowptr, colidx = _rand_csr(rng, 120)
    h = rng.standard_normal((120, 24))
    vals = rng.random(len(colidx))
    for v in (None, vals):
        a = sp.csr_matrix((np.ones(len(colidx)) if v is None else v, colidx, rowptr), shape=(120, 120))
        assert rel_err(O.spmm_csr(rowptr, colidx, v, h), a @ h) < 1e-13
        assert rel_err(O.spmm_csr_T(rowptr, colidx, v, h), a.T @ h) < 1e-13
        tr, tc, tv = O.csr_transpose(rowptr, colidx, v)
        assert rel_err(O.spmm_csr(tr, tc, tv, h), a.T @ h) < 1e-13


def test_gcn_filter_spektral_adds_identity_unconditionally():
    import scipy.sparse as sp
    rng = np.random.default_rng(1)
    rowptr, colidx = _rand_csr(rng, 40, 0.2, self_loops=True)
    v = O.gcn_filter_csr(rowptr, colidx, None, "spektral")
    a = sp.csr_matrix((np.ones(len(colidx)), colidx, rowptr), shape=(40, 40)).toarray()
    at = a + np.eye(40)                         # diagonal becomes 2
    d = at.sum(1) ** -0.5
    dense = d[:, None] * at * d[None, :]
    rows = np.repeat(np.arange(40), np.diff(rowptr))
    assert np.allclose(v, dense[rows, colidx], rtol=1e-13)
    assert np.allclose(O.gcn_filter_scipy(sp.csr_matrix(a)).toarray(), dense, rtol=1e-13)
    # PyG flavour keeps existing loops at weight 1
    vp = O.gcn_filter_csr(rowptr, colidx, None, "pyg")
    dp = a.sum(1) ** -0.5
    assert np.allclose(vp, (dp[:, None] * a * dp[None, :])[rows, colidx], rtol=1e-13)


def test_disjoint_collate_layout():
    from gcnx import synth
    graphs = synth.tiny_graphs(5, 8, seed=3)
    x, (idx, val, shape), i, y = O.disjoint_collate(graphs)
    n = sum(g[0].shape[0] for g in graphs)
    assert x.shape == (n, 8) and shape == (n, n) and y.shape == (5, 2)
    assert idx.dtype == np.int64 and i.dtype == np.int64
    key = idx[:, 0] * n + idx[:, 1]
    assert np.all(np.diff(key) > 0)                                   # row-major, unique
    assert np.array_equal(i, np.repeat(np.arange(5), [g[0].shape[0] for g in graphs]))
    assert np.all(i[idx[:, 0]] == i[idx[:, 1]])                       # block diagonal
    off = 0
    for g in graphs:                                                  # every block equals its graph
        m = g[0].shape[0]
        sel = (idx[:, 0] >= off) & (idx[:, 0] < off + m)
        dense = np.zeros((m, m)); dense[idx[sel, 0] - off, idx[sel, 1] - off] = val[sel]
        assert np.array_equal(dense, g[1].toarray())
        off += m


@pytest.mark.parametrize("cce_mode", ["logits", "probs"])
@pytest.mark.parametrize("pool", ["sum", "avg", "max"])
@pytest.mark.parametrize("weighted", [False, True])
def test_gcn2_gradients_match_torch_autograd(pool, weighted, cce_mode):
    import torch
    rng = np.random.default_rng(5)
    sizes = [7, 1, 12, 5]
    n = sum(sizes)
    gp = np.concatenate([[0], np.cumsum(sizes)])
    blocks = []
    import scipy.sparse as sp
    for s in sizes:
        rp, ci = _rand_csr(rng, s, 0.4)
        blocks.append(sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(s, s)))
    a = sp.block_diag(blocks).tocsr(); a.sort_indices()
    rowptr, colidx = a.indptr.astype(np.int64), a.indices.astype(np.int64)
    vals = O.gcn_filter_csr(rowptr, colidx, None) if weighted else None
    x = 0.3 * rng.standard_normal((n, 6))
    y = np.eye(2)[rng.integers(0, 2, len(sizes))]
    params = O.gcn2_init(rng, 6, 5, 2)
    for k in ("b1", "b2", "b3"):
        params[k] = 0.1 * rng.standard_normal(params[k].shape)
    loss, acc, grads, cache = O.gcn2_loss_and_grads(params, x, (rowptr, colidx, vals), gp, y, pool, cce_mode=cce_mode)

    rows = np.repeat(np.arange(n), np.diff(rowptr))
    at = torch.sparse_coo_tensor(np.stack([rows, colidx]),
                                 torch.tensor(np.ones(len(colidx)) if vals is None else vals), (n, n)).coalesce()
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    y1 = torch.relu(torch.sparse.mm(at, torch.tensor(x) @ tp["w1"]) + tp["b1"])
    y2 = torch.relu(torch.sparse.mm(at, y1 @ tp["w2"]) + tp["b2"])
    segs = [y2[gp[g]:gp[g + 1]] for g in range(len(sizes))]
    pooled = torch.stack([s.sum(0) if pool == "sum" else s.mean(0) if pool == "avg" else s.max(0).values for s in segs])
    logits = pooled @ tp["w3"] + tp["b3"]
    probs = torch.softmax(logits, 1)
    if cce_mode == "logits":     # tf.nn.softmax_cross_entropy_with_logits: what Keras' CCE runs inside tf.function
        tl = -(torch.tensor(y) * torch.log_softmax(logits, 1)).sum(1).mean()
    else:                        # the eager branch: clip, log
        tl = -(torch.tensor(y) * torch.log(torch.clamp(probs, 1e-7, 1 - 1e-7))).sum(1).mean()
    tl.backward()
    assert abs(loss - tl.item()) < 1e-12
    for k in grads:
        assert rel_err(grads[k], tp[k].grad.numpy()) < 1e-11, k
    # sharding identity: shard losses/grads normalised by the GLOBAL batch add up
    tot = {k: 0.0 for k in grads}; ltot = 0.0
    for g0, g1 in ((0, 2), (2, 4)):
        r0, r1 = gp[g0], gp[g1]
        e0, e1 = rowptr[r0], rowptr[r1]
        csr_s = (rowptr[r0:r1 + 1] - e0, colidx[e0:e1] - r0, None if vals is None else vals[e0:e1])
        l_s, _, g_s, _ = O.gcn2_loss_and_grads(params, x[r0:r1], csr_s, gp[g0:g1 + 1] - r0, y[g0:g1], pool, denom=len(sizes),
                                               cce_mode=cce_mode)
        ltot += l_s
        for k in g_s:
            tot[k] = tot[k] + g_s[k]
    assert abs(ltot - loss) < 1e-12
    for k in grads:
        assert rel_err(tot[k], grads[k]) < 1e-12, k


def test_finite_difference_gradcheck_gcnconv():
    rng = np.random.default_rng(9)
    rowptr, colidx = _rand_csr(rng, 15, 0.3)
    vals = O.gcn_filter_csr(rowptr, colidx, None)
    x = rng.standard_normal((15, 4)); w = rng.standard_normal((4, 3)); b = 0.1 * rng.standard_normal(3)
    r = rng.standard_normal((15, 3))

    def f(w_, b_, x_):
        return float((O.gcn_conv_fwd(x_, (rowptr, colidx, vals), w_, b_, "relu")[0] * r).sum())
    yv, cache = O.gcn_conv_fwd(x, (rowptr, colidx, vals), w, b, "relu")
    dx, dw, db = O.gcn_conv_bwd(r, cache, (rowptr, colidx, vals), w, "relu")
    eps = 1e-6
    for arr, grad, which in ((w, dw, 0), (b, db, 1), (x, dx, 2)):
        num = np.zeros_like(arr)
        it = np.nditer(arr, flags=["multi_index"])
        for _ in it:
            i = it.multi_index
            p, m = arr.copy(), arr.copy(); p[i] += eps; m[i] -= eps
            args = [w, b, x]; args[which] = p; fp = f(*args); args[which] = m; fm = f(*args)
            num[i] = (fp - fm) / (2 * eps)
        assert rel_err(grad, num) < 1e-6


@pytest.mark.parametrize("aggregate,pool", [("sum", "sum"), ("mean", "avg"), ("sum", "max"), ("mean", "sum"), ("max", "sum"), ("min", "avg"),
                                            ("prod", "sum")])
def test_general_gnn_gradients_match_torch_autograd(aggregate, pool):
    """n1 tier (GeneralGNN-complete): BN(train) + PReLU + concat-skip + sum-aggregation (gcn.py:320's defaults), and the
    Spektral options aggregate="mean" / pool="avg" | "max" (r3)."""
    import torch
    rng = np.random.default_rng(2)
    sizes = [6, 9, 4]
    n = sum(sizes); gp = np.concatenate([[0], np.cumsum(sizes)])
    import scipy.sparse as sp
    blocks = []
    for s in sizes:
        rp, ci = _rand_csr(rng, s, 0.4)
        blocks.append(sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(s, s)))
    a = sp.block_diag(blocks).tocsr(); a.sort_indices()
    rowptr, colidx = a.indptr.astype(np.int64), a.indices.astype(np.int64)
    x = rng.standard_normal((n, 5)); y = np.eye(2)[rng.integers(0, 2, 3)]
    layers = O.general_gnn_init(rng, 5, 2, hidden=8, message_passing=2, pre=2, post=2)
    for grp in layers.values():
        for p in grp:
            if "alpha" in p:
                p["alpha"] = 0.25 * rng.random(p["alpha"].shape)
            p["gamma"] = 1 + 0.1 * rng.standard_normal(p["gamma"].shape)
            p["beta"] = 0.1 * rng.standard_normal(p["beta"].shape)
            p["bias"] = 0.1 * rng.standard_normal(p["bias"].shape)
    loss, acc, grads, probs, stats = O.general_gnn_loss_and_grads(layers, x, (rowptr, colidx, None), gp, y, aggregate=aggregate, pool=pool)

    T = lambda v: torch.tensor(v, requires_grad=True)
    tl = {g: [{k: T(v) for k, v in p.items() if not k.startswith("moving")} for p in ps] for g, ps in layers.items()}
    dense = a.toarray()
    if aggregate == "mean":
        deg = dense.sum(1, keepdims=True)
        dense = np.where(deg > 0, dense / np.maximum(deg, 1), 0.0)
    at = torch.tensor(dense)

    def block(h, p, final=False):
        z = h @ p["kernel"] + p["bias"]
        mu, var = z.mean(0), z.var(0, unbiased=False)
        zb = p["gamma"] * (z - mu) / torch.sqrt(var + 1e-3) + p["beta"]
        if final:
            return torch.softmax(zb, 1)
        return torch.relu(zb) + p["alpha"] * torch.minimum(zb, torch.zeros_like(zb))
    out = torch.tensor(x)
    for p in tl["pre"]:
        out = block(out, p)
    nbrs = [colidx[rowptr[t]:rowptr[t + 1]] for t in range(n)]

    def aggregate_t(h):          # "max" / "min": per target row over its messages (torch.amax / amin share the gradient among ties)
        if aggregate in ("sum", "mean"):
            return at @ h
        if aggregate == "prod":  # tf.math.unsorted_segment_prod (r4)
            return torch.stack([h[torch.tensor(ix)].prod(0) for ix in nbrs])
        f = torch.amax if aggregate == "max" else torch.amin
        return torch.stack([f(h[torch.tensor(ix)], 0) for ix in nbrs])
    for p in tl["gnn"]:
        out = torch.cat([aggregate_t(block(out, p)), out], 1)
    red = {"sum": lambda t: t.sum(0), "avg": lambda t: t.mean(0), "max": lambda t: t.amax(0)}[pool]
    out = torch.stack([red(out[gp[g]:gp[g + 1]]) for g in range(3)])
    out = block(out, tl["post"][0]); pr = block(out, tl["post"][1], final=True)
    tloss = -(torch.tensor(y) * torch.log(torch.clamp(pr, 1e-7, 1 - 1e-7))).sum(1).mean()
    tloss.backward()
    assert abs(loss - tloss.item()) < 1e-12 and rel_err(probs, pr.detach().numpy()) < 1e-12
    for grp in grads:
        for k, g in enumerate(grads[grp]):
            for name, val in g.items():
                ref = tl[grp][k][name].grad.numpy()   # Dense bias under BN has an exactly-zero gradient
                assert np.allclose(val, ref, rtol=1e-9, atol=1e-13), (grp, k, name)


def test_aggregate_minmax_ties_share_the_gradient():
    """aggregate="max" / "min" with exact ties (integer-valued messages): value, tie count and TensorFlow's gradient rule --
    every message equal to the extremum gets dy / count -- against torch.amax / amin, whose backward shares the gradient the
    same way; an empty row gets the lowest / largest float32 and passes no gradient."""
    import torch
    rng = np.random.default_rng(8)
    rowptr, colidx = _rand_csr(rng, 12, 0.35)
    rowptr = rowptr.copy(); colidx = colidx.copy()
    e0, e1 = rowptr[5], rowptr[6]                      # make row 5 empty
    colidx = np.concatenate([colidx[:e0], colidx[e1:]]); rowptr[6:] -= (e1 - e0)
    h = rng.integers(-2, 3, (12, 5)).astype(np.float64)
    dy = rng.standard_normal((12, 5))
    for mode in ("max", "min"):
        out, cnt = O.aggregate_minmax(rowptr, colidx, h, mode)
        dh = O.aggregate_minmax_bwd(rowptr, colidx, h, out, cnt, dy)
        assert out[5, 0] == (-O.F32_MAX if mode == "max" else O.F32_MAX) and cnt[5, 0] == 0
        assert cnt.max() >= 2                            # ties are present
        th = torch.tensor(h, requires_grad=True)
        f = torch.amax if mode == "max" else torch.amin
        rows = [t for t in range(12) if rowptr[t + 1] > rowptr[t]]
        tout = torch.stack([f(th[torch.tensor(colidx[rowptr[t]:rowptr[t + 1]])], 0) for t in rows])
        (tout * torch.tensor(dy[rows])).sum().backward()
        assert np.array_equal(out[rows], tout.detach().numpy())
        assert np.allclose(dh, th.grad.numpy(), rtol=1e-12, atol=1e-14)


def test_aggregate_prod_follows_tensorflows_zero_aware_gradient():
    """aggregate="prod" (r4) with exact zeros among the messages (integer values): the product, and TensorFlow's
    _UnsortedSegmentProdGrad -- prod / message for a non-zero message, the product of the others for the ONLY zero of a row, nothing
    where a row holds two or more zeros -- against torch.prod, whose backward handles zeros the same way; a row without messages is 1."""
    import torch
    rng = np.random.default_rng(11)
    rowptr, colidx = _rand_csr(rng, 14, 0.3)
    rowptr = rowptr.copy(); colidx = colidx.copy()
    e0, e1 = rowptr[3], rowptr[4]                      # make row 3 empty
    colidx = np.concatenate([colidx[:e0], colidx[e1:]]); rowptr[4:] -= (e1 - e0)
    h = rng.integers(-2, 3, (14, 6)).astype(np.float64)
    dy = rng.standard_normal((14, 6))
    out, aux = O.aggregate_prod(rowptr, colidx, h)
    dh = O.aggregate_prod_bwd(rowptr, colidx, h, out, aux, dy)
    zeros = np.array([[(h[colidx[rowptr[t]:rowptr[t + 1]], c] == 0).sum() for c in range(6)] for t in range(14)])
    assert (zeros == 0).any() and (zeros == 1).any() and (zeros >= 2).any()          # all three branches of the rule occur
    assert np.all(out[3] == 1.0)
    th = torch.tensor(h, requires_grad=True)
    tout = torch.stack([th[torch.tensor(colidx[rowptr[t]:rowptr[t + 1]])].prod(0) for t in range(14)])
    (tout * torch.tensor(dy)).sum().backward()
    assert np.array_equal(out, tout.detach().numpy())
    assert np.allclose(dh, th.grad.numpy(), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("connectivity,batch_norm,act,rate", [("sum", True, "prelu", 0.0), ("cat", False, "prelu", 0.0),
                                                              ("cat", True, "relu", 0.0), ("cat", True, "prelu", 0.4),
                                                              ("sum", False, None, 0.25), ("sum", True, "relu", 0.5)])
def test_general_gnn_options_match_torch_autograd(connectivity, batch_norm, act, rate):
    """Spektral's GeneralGNN options beside gcn.py:320's defaults as the oracle restates them (r3): connectivity="sum",
    batch_norm=False, hidden_activation "relu" / None, dropout > 0 (the Dropout layer sits between BatchNormalization and the
    activation in every MLP / GeneralConv layer, the last post layer included; its factors are given, keep / (1 - rate)) --
    loss, probabilities and every gradient against torch autograd on the same graph."""
    import torch
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    sizes = [7, 5, 8]
    n = sum(sizes); gp = np.concatenate([[0], np.cumsum(sizes)])
    blocks = []
    for sz in sizes:
        rp, ci = _rand_csr(rng, sz, 0.4)
        blocks.append(sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(sz, sz)))
    a = sp.block_diag(blocks).tocsr(); a.sort_indices()
    rowptr, colidx = a.indptr.astype(np.int64), a.indices.astype(np.int64)
    x = rng.standard_normal((n, 5)); y = np.eye(2)[rng.integers(0, 2, 3)]
    hid = 8
    layers = O.general_gnn_init(rng, 5, 2, hidden=hid, message_passing=2, pre=2, post=2, connectivity=connectivity,
                                batch_norm=batch_norm, hidden_activation=act)
    assert layers["gnn"][1]["kernel"].shape[0] == (2 * hid if connectivity == "cat" else hid)
    assert ("gamma" in layers["pre"][0]) == batch_norm and ("alpha" in layers["pre"][0]) == (act == "prelu")
    for grp in layers.values():
        for p in grp:
            if "alpha" in p:
                p["alpha"] = 0.25 * rng.random(p["alpha"].shape)
            if "gamma" in p:
                p["gamma"] = 1 + 0.1 * rng.standard_normal(p["gamma"].shape)
                p["beta"] = 0.1 * rng.standard_normal(p["beta"].shape)
            p["bias"] = 0.1 * rng.standard_normal(p["bias"].shape)
    drops = None
    if rate > 0:
        mk = lambda rows, p: (rng.random((rows, p["kernel"].shape[1])) >= rate) / (1.0 - rate)
        drops = {"pre": [mk(n, p) for p in layers["pre"]], "gnn": [mk(n, p) for p in layers["gnn"]],
                 "post": [mk(3, p) for p in layers["post"]]}
    kw = dict(connectivity=connectivity, hidden_activation=act, drops=drops)
    loss, acc, grads, probs, stats = O.general_gnn_loss_and_grads(layers, x, (rowptr, colidx, None), gp, y, cce_mode="probs", **kw)
    # inference ignores the Dropout layers
    p_inf, _, _ = O.general_gnn_forward(layers, x, (rowptr, colidx, None), gp, False, **kw)
    p_inf0, _, _ = O.general_gnn_forward(layers, x, (rowptr, colidx, None), gp, False, connectivity=connectivity, hidden_activation=act)
    assert np.array_equal(p_inf, p_inf0)

    T = lambda v: torch.tensor(v, requires_grad=True)
    tl = {g: [{k: T(v) for k, v in p.items() if not k.startswith("moving")} for p in ps] for g, ps in layers.items()}
    at = torch.tensor(a.toarray())

    def block(h, p, drop, final=False):
        z = h @ p["kernel"] + p["bias"]
        if "gamma" in p:
            mu, var = z.mean(0), z.var(0, unbiased=False)
            z = p["gamma"] * (z - mu) / torch.sqrt(var + 1e-3) + p["beta"]
        if drop is not None:
            z = z * torch.tensor(drop)
        if final:
            return torch.softmax(z, 1)
        if act == "prelu":
            return torch.relu(z) + p["alpha"] * torch.minimum(z, torch.zeros_like(z))
        return torch.relu(z) if act == "relu" else z
    dr = lambda grp, k: None if drops is None else drops[grp][k]
    out = torch.tensor(x)
    for k, p in enumerate(tl["pre"]):
        out = block(out, p, dr("pre", k))
    for k, p in enumerate(tl["gnn"]):
        z = at @ block(out, p, dr("gnn", k))
        out = torch.cat([z, out], 1) if connectivity == "cat" else z + out
    out = torch.stack([out[gp[g]:gp[g + 1]].sum(0) for g in range(3)])
    out = block(out, tl["post"][0], dr("post", 0)); pr = block(out, tl["post"][1], dr("post", 1), final=True)
    tloss = -(torch.tensor(y) * torch.log(torch.clamp(pr, 1e-7, 1 - 1e-7))).sum(1).mean()
    tloss.backward()
    assert abs(loss - tloss.item()) < 1e-12 and rel_err(probs, pr.detach().numpy()) < 1e-12
    for grp in grads:
        for k, g in enumerate(grads[grp]):
            assert set(g) == set(tl[grp][k]), (grp, k)
            for name, val in g.items():
                ref = tl[grp][k][name].grad.numpy()
                assert np.allclose(val, ref, rtol=1e-9, atol=1e-13), (grp, k, name)


@pytest.mark.parametrize("name", GOLDEN)
def test_numpy_oracle_reproduces_goldens(name):
    g = load_golden(name)
    hb = golden_batch(g)
    params = {k: g["p_" + k].astype(np.float64) for k in O.GCN2_PARAM_ORDER}
    vals = None if hb.vals is None else hb.vals.astype(np.float64)
    for mode, lk, gk in (("logits", "loss", "g_"), ("probs", "loss_probs", "gp_")):
        loss, acc, grads, cache = O.gcn2_loss_and_grads(params, hb.x.astype(np.float64),
                                                        (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals),
                                                        hb.graph_ptr, hb.y.astype(np.float64), str(g["pool"]), cce_mode=mode)
        assert abs(loss - float(g[lk])) < 1e-12 * max(1.0, abs(loss))
        assert acc == float(g["acc"])
        assert rel_err(cache["probs"], g["probs"]) < 1e-12
        assert rel_err(cache["y2"], g["y2"]) < 1e-6          # goldens store activations in fp32
        for k in O.GCN2_PARAM_ORDER:
            assert rel_err(grads[k], g[gk + k]) < 1e-6 or not np.any(g[gk + k]), (mode, k)


@pytest.mark.parametrize("name", [n for n in GOLDEN if "max" not in n and "avg" not in n])
def test_c_oracle_matches_goldens_fp32(name):
    """The fp32 C restatement (sum pool, symmetric adjacency) against the fp64 goldens at the
    1e-4 bar the GPU path is held to."""
    from oracle import c_oracle
    g = load_golden(name)
    hb = golden_batch(g)
    flat = np.concatenate([g["p_" + k].ravel() for k in O.GCN2_PARAM_ORDER]).astype(np.float32)
    for mode, lk, gk in (("logits", "loss", "g_"), ("probs", "loss_probs", "gp_")):
        m = c_oracle.Gcn2Cpu(hb, g["p_w1"].shape[1], 2, flat)
        loss, acc = m.step(lr=0.0, cce=mode)
        assert abs(loss - float(g[lk])) < 1e-4 * max(1.0, abs(float(g[lk])))
        assert acc == pytest.approx(float(g["acc"]))
        off = 0
        for k in O.GCN2_PARAM_ORDER:
            n = g[gk + k].size
            got = m.grads[off:off + n].reshape(g[gk + k].shape)
            assert rel_err(got, g[gk + k]) < 1e-4 or (not np.any(g[gk + k]) and not np.any(got)), (mode, k)
            off += n
    # SGD apply (gcn.py:338)
    m2 = c_oracle.Gcn2Cpu(hb, g["p_w1"].shape[1], 2, flat)
    m2.step(lr=float(g["lr"]))
    assert np.allclose(m2.params, flat - np.float32(g["lr"]) * m2.grads, rtol=1e-6, atol=1e-7)


def test_c_oracle_kernels_match_numpy():
    from oracle import c_oracle
    rng = np.random.default_rng(4)
    rowptr, colidx = _rand_csr(rng, 200, 0.05)
    vals = O.gcn_filter_csr(rowptr, colidx, None).astype(np.float32)
    h = rng.standard_normal((200, 20)).astype(np.float32)
    b = rng.standard_normal(20).astype(np.float32)
    ref = np.maximum(O.spmm_csr(rowptr, colidx, vals.astype(np.float64), h.astype(np.float64)) + b, 0)
    out = c_oracle.spmm_csr(rowptr.astype(np.int32), colidx.astype(np.int32), vals, h, b, relu=True)
    assert rel_err(out, ref) < 1e-6
    w = rng.standard_normal((20, 12)).astype(np.float32)
    assert rel_err(c_oracle.gemm(h, w), h.astype(np.float64) @ w.astype(np.float64)) < 1e-6


def test_lr_schedule_and_losses():
    # PiecewiseConstantDecay indexed by optimizer step, epochs=5 -> boundaries [0, 1] (gcn.py:321-324)
    assert [O.piecewise_lr(s, 5) for s in (0, 1, 2, 100)] == [0.02, 0.002, 0.0002, 0.0002]
    p = np.array([[1.0, 0.0], [0.5, 0.5]]); y = np.array([[1.0, 0.0], [0.0, 1.0]])
    assert np.isclose(O.cce_loss(y, p), (-np.log(1 - 1e-7) - np.log(0.5)) / 2)      # clip at 1-1e-7
    assert O.categorical_accuracy(y, np.array([[0.9, 0.1], [0.8, 0.2]])) == 0.5
    # the two Keras code paths: identical while no probability saturates, different beyond |dz| ~ 16.1
    z = np.array([[2.0, -1.0], [0.3, 0.1]])
    assert np.isclose(O.cce_loss_from_logits(y, z), O.cce_loss(y, O.softmax(z)), rtol=1e-12)
    assert np.allclose(O.softmax_cce_grad_from_logits(y, O.softmax(z)), O.softmax_cce_grad(y, O.softmax(z)), atol=1e-15)
    zs = np.array([[-30.0, 30.0], [0.3, 0.1]])                 # graph 0: true class at p = e^-60
    assert np.isclose(O.cce_loss_from_logits(y, zs), (60.0 + np.log1p(np.exp(-60.0)) + np.log(1 + np.exp(0.2))) / 2)
    assert np.isclose(O.cce_loss(y, O.softmax(zs)), (-np.log(1e-7) + np.log(1 + np.exp(0.2))) / 2)
    assert np.all(O.softmax_cce_grad(y, O.softmax(zs))[0] == 0) and abs(O.softmax_cce_grad_from_logits(y, O.softmax(zs))[0, 0] + 0.5) < 1e-12


def test_bf16_rounding_is_nearest_even():
    """bf16_bits against a scalar restatement (struct): ties go to the even mantissa, everything else to the nearest."""
    import struct
    from oracle import gcn_oracle as o

    def ref(x):
        u = struct.unpack("<I", struct.pack("<f", x))[0]
        lower, rest = u >> 16, u & 0xffff
        if rest > 0x8000 or (rest == 0x8000 and (lower & 1)):
            lower += 1
        return lower & 0xffff

    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(2000).astype(np.float32) * 10.0 ** rng.integers(-6, 6, 2000),
                         np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -1.0 - 2.0 ** -8, 0.0, 2.0 ** -126, 65504.0], np.float32)])
    assert [int(b) for b in o.bf16_bits(xs)] == [ref(float(x)) for x in xs]
    assert o.bf16_bits(np.float32([1.0 + 2.0 ** -8]))[0] == 0x3f80 and o.bf16_bits(np.float32([1.0 + 3 * 2.0 ** -8]))[0] == 0x3f82
    back = o.bf16_from_bits(o.bf16_bits(xs))
    assert np.all(np.abs(back - xs) <= np.abs(xs) * 2.0 ** -8)


def test_general_gnn_backward_on_a_given_side_of_the_activation_kinks():
    """general_gnn_loss_and_grads(masks=...) (test infrastructure for the GPU parity tests, r4): with every layer's own mask
    zb > 0 it is the plain backward; flipping ONE entry of one hidden layer's mask changes that layer's alpha gradient by
    exactly the entry's term and leaves the loss alone (the forward pass does not depend on the masks)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(12)
    graphs = []
    for _ in range(3):
        n = int(rng.integers(6, 12))
        a = (rng.random((n, n)) < 0.3).astype(np.float64)
        a = np.maximum(a, a.T); np.fill_diagonal(a, 1.0)
        graphs.append((rng.standard_normal((n, 5)), sp.csr_matrix(a), np.eye(2)[rng.integers(0, 2)]))
    x, (idx, val, shape), i, y = O.disjoint_collate(graphs)
    rp, ci = O.coo_to_csr(idx, shape[0])
    gp = O.graph_ptr_from_ids(i, 3)
    layers = O.general_gnn_init(rng, 5, 2, hidden=8, message_passing=2)
    for grp in layers.values():
        for p in grp:
            if "alpha" in p:
                p["alpha"] = 0.25 * rng.random(p["alpha"].shape)
    csr = (rp, ci, None)
    l0, a0, g0, p0, _ = O.general_gnn_loss_and_grads(layers, x, csr, gp, y)
    _, caches, _ = O.general_gnn_forward(layers, x, csr, gp, True)
    own = {grp: [(c["zb"] > 0) if k < len(layers[grp]) - (grp == "post") else None for k, c in enumerate(caches[grp])]
           for grp in ("pre", "gnn", "post")}
    l1, a1, g1, p1, _ = O.general_gnn_loss_and_grads(layers, x, csr, gp, y, masks=own)
    assert l1 == l0 and a1 == a0 and np.array_equal(p1, p0)
    for grp in g0:
        for ga, gb in zip(g0[grp], g1[grp]):
            for k in ga:
                assert np.array_equal(ga[k], gb[k]), (grp, k)
    flipped = {grp: [None if m is None else m.copy() for m in own[grp]] for grp in own}
    r, c = 2, 3
    flipped["gnn"][1][r, c] ^= True
    l2, _, g2, _, _ = O.general_gnn_loss_and_grads(layers, x, csr, gp, y, masks=flipped)
    assert l2 == l0
    d_alpha = g2["gnn"][1]["alpha"] - g0["gnn"][1]["alpha"]
    assert np.count_nonzero(d_alpha) == 1 and d_alpha[c] != 0
    assert not np.array_equal(g2["gnn"][1]["kernel"], g0["gnn"][1]["kernel"])
    assert all(np.array_equal(a_[k], b_[k]) for a_, b_ in zip(g2["post"], g0["post"]) for k in a_)    # downstream of nothing
