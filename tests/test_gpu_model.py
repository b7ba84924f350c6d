"""GPU parity of the whole path (loader -> model forward -> loss -> gradients -> SGD) against the
committed golden vectors and the CPU oracle, through the Spektral-shaped host surface."""
import numpy as np
import pytest

from conftest import GOLDEN, golden_batch, load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
ORDER = ("w1", "b1", "w2", "b2", "w3", "b3")


def _model_from_golden(ctx, g, use_graph=False):
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2
    hb = golden_batch(g)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    m = GCN2(ctx, 2, hidden=g["p_w1"].shape[1], pool=str(g["pool"]), use_graph=use_graph)
    m.build(hb.f)
    m.set_weights([g["p_" + k] for k in ORDER])
    return m, batch, hb


@pytest.mark.parametrize("name", GOLDEN)
def test_gcn2_matches_golden_vectors(ctx, name):
    g = load_golden(name)
    m, batch, hb = _model_from_golden(ctx, g)
    m.loss_and_grads(batch, None)
    loss, acc = m.fetch_metrics(hb.n_graphs)
    assert abs(loss - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    assert acc == pytest.approx(float(g["acc"]))
    assert rel_err(m._bufs["y2"].numpy(), g["y2"]) < TOL
    assert rel_err(m._bufs["pooled"].numpy(), g["pooled"]) < TOL
    assert rel_err(m._bufs["probs"].numpy(), g["probs"]) < TOL
    grads = m.gradients()
    for k in ORDER:
        assert rel_err(grads[k], g["g_" + k]) < TOL, k
    # optimiser apply (gcn.py:338): w <- w - lr g
    before = m.get_weights()
    m.train_step(batch, None, lr=float(g["lr"]))
    for k, w0, w1 in zip(ORDER, before, m.get_weights()):
        assert np.allclose(w1, w0 - np.float32(g["lr"]) * grads[k], rtol=0, atol=1e-6), k
    # forward-only surface: model(inputs, training=False) -> probabilities
    m.set_weights([g["p_" + k] for k in ORDER])
    assert rel_err(m(batch, training=False), g["probs"]) < TOL


@pytest.mark.parametrize("name", [n for n in GOLDEN if "tiny_weighted" in n or "f128" in n or "ecoli" in n])
def test_gcn2_bf16x3_gemm_meets_the_fp32_bar(ctx, name):
    """Whole step with the bf16x3 MFMA GEMMs against the fp64 golden vectors at the same 1e-4."""
    g = load_golden(name)
    m, batch, hb = _model_from_golden(ctx, g)
    m.prec = "bf16x3"
    m.loss_and_grads(batch, None)
    loss, acc = m.fetch_metrics(hb.n_graphs)
    assert abs(loss - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    assert rel_err(m._bufs["probs"].numpy(), g["probs"]) < TOL
    for k, gk in m.gradients().items():
        assert rel_err(gk, g["g_" + k]) < TOL, k


def test_disjoint_loader_to_model_end_to_end(ctx):
    """config 1 plumbing: Graph objects -> DisjointLoader -> ((x, a, i), y) -> device COO->CSR ->
    device gcn_filter -> model, vs the oracle fed by its own collate."""
    from oracle import gcn_oracle as O
    from gcnx import DisjointLoader, Graph, ListDataset, synth
    from gcnx.models import DeviceBatch, GCN2
    raw = synth.tiny_graphs(16, 32, seed=0)
    loader = DisjointLoader(ListDataset([Graph(x=x, a=a, y=y) for x, a, y in raw]), batch_size=16, epochs=1, shuffle=False)
    inputs, target = next(loader)
    batch = DeviceBatch.from_host(ctx, inputs, target, normalize="spektral")
    m = GCN2(ctx, 2, hidden=32, use_graph=False, seed=3)
    loss, acc = m.train_step(batch, None, lr=0.0)
    w = dict(zip(ORDER, m.get_weights()))
    x, (idx, val, shape), i, y = O.disjoint_collate(raw)
    rowptr, colidx = O.coo_to_csr(idx, shape[0])
    vals = O.gcn_filter_csr(rowptr, colidx, None)
    params = {k: v.astype(np.float64) for k, v in w.items()}
    rl, ra, rg, cache = O.gcn2_loss_and_grads(params, x.astype(np.float32).astype(np.float64), (rowptr, colidx, vals),
                                              O.graph_ptr_from_ids(i, 16), y.astype(np.float64))
    assert abs(loss - rl) < TOL * max(1, rl) and acc == pytest.approx(ra)
    for k, gk in m.gradients().items():
        assert rel_err(gk, rg[k]) < TOL, k


def test_layer_surface_gcnconv_pool_dense(ctx):
    """The Spektral call surface: GCNConv([x, a]), GlobalSumPool([x, i]), Dense(x) + backward."""
    from oracle import gcn_oracle as O
    from gcnx import synth
    from gcnx.device import DeviceCSR
    from gcnx.layers import Dense, GCNConv, GlobalMaxPool, GlobalSumPool
    hb = synth.ecoli_batch(3, 16, seed=9)
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
    x = ctx.to_device(hb.x)
    conv = GCNConv(24, activation="relu", seed=1)
    y = conv([x, a])
    w, b = conv.get_weights()
    b = (0.1 * np.random.default_rng(0).standard_normal(24)).astype(np.float32)
    conv.set_weights([w, b]); y = conv([x, a])
    csr = (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals.astype(np.float64))
    ry, cache = O.gcn_conv_fwd(hb.x.astype(np.float64), csr, w.astype(np.float64), b.astype(np.float64), "relu")
    assert rel_err(y.numpy(), ry) < TOL
    for Pool, mode in ((GlobalSumPool, "sum"), (GlobalMaxPool, "max")):
        pool = Pool()
        p = pool([y, hb.ids()])                      # accepts the loader's id vector i
        rp, rarg = O.global_pool_fwd(ry, hb.graph_ptr, mode)
        assert rel_err(p.numpy(), rp) < TOL
    dense = Dense(2, seed=2)
    logits = dense(p)
    wd, bd = dense.get_weights()
    assert rel_err(logits.numpy(), rp @ wd.astype(np.float64) + bd) < TOL
    # backward chain
    dl = np.random.default_rng(1).standard_normal(logits.shape).astype(np.float32)
    dp = dense.backward(ctx.to_device(dl))
    assert rel_err(dense.grads["kernel"].numpy(), rp.T @ dl.astype(np.float64)) < TOL
    dyv = pool.backward(dp)
    dxv = conv.backward(dyv)
    rdp = dl.astype(np.float64) @ wd.astype(np.float64).T
    rdy = O.global_pool_bwd(rdp, hb.graph_ptr, hb.n, "max", rarg)
    rdx, rdw, rdb = O.gcn_conv_bwd(rdy, cache, csr, w.astype(np.float64), "relu")
    assert rel_err(conv.grads["kernel"].numpy(), rdw) < TOL and rel_err(conv.grads["bias"].numpy(), rdb) < TOL
    assert rel_err(dxv.numpy(), rdx) < TOL


def test_training_trajectory_and_graph_replay(ctx):
    """5 SGD steps on one batch: HIP-graph replay == eager bitwise, and both follow the fp32 CPU
    restatement of train_step (gcn.py:330-340) within 1e-4."""
    from oracle import c_oracle
    g = load_golden("gcn2_cfg1_tiny_weighted")
    runs = []
    for use_graph in (False, True):
        m, batch, hb = _model_from_golden(ctx, g, use_graph)
        hist = [m.train_step(batch, None, lr=0.02) for _ in range(5)]
        runs.append((hist, m.get_weights()))
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a, b)
    assert runs[0][0] == runs[1][0]
    flat = np.concatenate([g["p_" + k].ravel() for k in ORDER]).astype(np.float32)
    cpu = c_oracle.Gcn2Cpu(golden_batch(g), 32, 2, flat)
    ref_hist = [cpu.step(lr=0.02) for _ in range(5)]
    for (l, a), (rl, ra) in zip(runs[0][0], ref_hist):
        assert abs(l - rl) < TOL * max(1, rl) and a == pytest.approx(ra)
    got = np.concatenate([w.ravel() for w in runs[0][1]])
    assert rel_err(got, cpu.params) < TOL
    assert ref_hist[-1][0] != ref_hist[0][0]         # the weights moved


def test_ecoli_config2_full_size_vs_cpu_restatement(ctx):
    """BASELINE config 2 (B=32, F=128, fp32) at full size against the fp32 C restatement."""
    from oracle import c_oracle
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2
    hb = synth.ecoli_batch()
    hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    m = GCN2(ctx, 2, hidden=128, seed=0)
    loss, acc = m.train_step(batch, None, lr=0.0)
    flat = np.concatenate([w.ravel() for w in m.get_weights()])
    cpu = c_oracle.Gcn2Cpu(hb, 128, 2, flat)
    rl, ra = cpu.step(lr=0.0)
    assert abs(loss - rl) < TOL * max(1, rl) and acc == pytest.approx(ra)
    got = np.concatenate([m.gradients()[k].ravel() for k in ORDER])
    assert rel_err(got, cpu.grads) < TOL
    # run-to-run determinism (atomics-free reductions)
    m.loss_and_grads(batch, None)
    again = np.concatenate([m.gradients()[k].ravel() for k in ORDER])
    assert np.array_equal(got, again)
