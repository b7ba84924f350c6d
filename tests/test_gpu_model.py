"""GPU parity of the whole path (loader -> model forward -> loss -> gradients -> SGD) against the
committed golden vectors and the CPU oracle, through the Spektral-shaped host surface."""
import numpy as np
import pytest

from gcnx.models import GCN2

from conftest import GOLDEN, PARITY_REPORT as PARITY_NOTES, assert_close, golden_batch, load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
ORDER = ("w1", "b1", "w2", "b2", "w3", "b3")


def _model_from_golden(ctx, g, use_graph=False, **kw):
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2
    hb = golden_batch(g)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    m = GCN2(ctx, 2, hidden=g["p_w1"].shape[1], pool=str(g["pool"]), use_graph=use_graph, **kw)
    m.build(hb.f)
    m.set_weights([g["p_" + k] for k in ORDER])
    return m, batch, hb


def _grad_ok(got, ref, tol=TOL):
    """1e-4 of the largest reference entry; a reference gradient that is identically zero (every graph clipped in the
    eager CCE form) must come back as zeros."""
    if not np.any(ref):
        return not np.any(got)
    assert_close(got, ref, tol, "gradient vs golden / oracle")     # both readings of the bar, kept for the parity report
    return True


@pytest.mark.parametrize("cce", ["logits", "probs"])
@pytest.mark.parametrize("name", GOLDEN)
def test_gcn2_matches_golden_vectors(ctx, name, cce):
    """cce = "logits": the loss train_step computes under tf.function (golden keys loss, g_*; the model's default);
    "probs": the eager renormalise-and-clip form (keys loss_probs, gp_*).  The gcn2_saturated_* fixtures are where the
    two differ."""
    g = load_golden(name)
    lk, gk = ("loss", "g_") if cce == "logits" else ("loss_probs", "gp_")
    m, batch, hb = _model_from_golden(ctx, g, cce_train=cce)
    if cce == "logits":
        assert GCN2(ctx, 2).cce_train == "logits" and GCN2(ctx, 2).cce_eval == "probs"      # the defaults
    m.loss_and_grads(batch, None)
    loss, acc = m.fetch_metrics(hb.n_graphs)
    assert abs(loss - float(g[lk])) < TOL * max(1.0, abs(float(g[lk])))
    assert acc == pytest.approx(float(g["acc"]))
    assert rel_err(m._bufs["y2"].numpy(), g["y2"]) < TOL
    assert rel_err(m._bufs["pooled"].numpy(), g["pooled"]) < TOL
    assert rel_err(m._bufs["probs"].numpy(), g["probs"]) < TOL
    grads = m.gradients()
    for k in ORDER:
        assert _grad_ok(grads[k], g[gk + k]), k
    # evaluate() semantics (eager, gcn.py:351-354): loss by cce_eval, whatever the training form
    el, ea, _ = m.evaluate_batch(batch, None)
    assert abs(el - float(g["loss_probs"])) < TOL * max(1.0, abs(float(g["loss_probs"]))) and ea == pytest.approx(float(g["acc"]))
    m.cce_eval = "logits"                                  # TF >= 2.6: the eager output carries _keras_logits
    el, _, _ = m.evaluate_batch(batch, None)
    assert abs(el - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    m.loss_and_grads(batch, None)
    # optimiser apply (gcn.py:338): w <- w - lr g
    before = m.get_weights()
    m.train_step(batch, None, lr=float(g["lr"]))
    for k, w0, w1 in zip(ORDER, before, m.get_weights()):
        assert np.allclose(w1, w0 - np.float32(g["lr"]) * grads[k], rtol=0, atol=1e-6), k
    # forward-only surface: model(inputs, training=False) -> probabilities
    m.set_weights([g["p_" + k] for k in ORDER])
    assert rel_err(m(batch, training=False), g["probs"]) < TOL


@pytest.mark.parametrize("name", [n for n in GOLDEN if "f128" in n or "ecoli" in n][:2])
def test_gcn2_folded_pool_backward_equals_the_unfused_chain(ctx, name, monkeypatch):
    """Small batches fold pool' and the ReLU mask into the backward aggregation (gcnx_spmm_csr_pool_bwd); with
    GCNX_FOLD=0 the materialised chain runs.  Both must meet the golden gradients and agree with each other."""
    g = load_golden(name)
    got = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("GCNX_FOLD", fold)
        m, batch, hb = _model_from_golden(ctx, g)
        m.loss_and_grads(batch, None)
        got[fold] = m.gradients()
        for k in ORDER:
            assert rel_err(got[fold][k], g["g_" + k]) < TOL, (fold, k)
    for k in ORDER:
        assert rel_err(got["1"][k], got["0"][k]) < 2e-5, k


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


def test_device_side_collate_equals_host_loader(ctx):
    """DeviceDataset + DeviceDisjointLoader assemble the same batches as DisjointLoader +
    DeviceBatch.from_host (same seed, same order): features, CSR (row pointers / column indices exact), filter
    values, labels, graph pointers -- including a ragged last batch -- and the model takes the same step on them."""
    from gcnx import DisjointLoader, Graph, ListDataset, synth, DeviceDataset, DeviceDisjointLoader
    from gcnx.models import DeviceBatch, GCN2
    raw = synth.tiny_graphs(11, 16, seed=4)
    ds = ListDataset([Graph(x=x, a=a, y=y) for x, a, y in raw])
    host = DisjointLoader(ds, batch_size=4, epochs=2, shuffle=True, seed=9)
    dds = DeviceDataset(ctx, ds, normalize="spektral")
    dev = DeviceDisjointLoader(dds, batch_size=4, epochs=2, shuffle=True, seed=9)
    assert dev.steps_per_epoch == host.steps_per_epoch == 3
    m1, m2 = GCN2(ctx, 2, hidden=16, use_graph=False, seed=1), GCN2(ctx, 2, hidden=16, use_graph=False, seed=1)
    n_batches = 0
    for (inputs, target), (dbatch, _) in zip(host, dev):
        hbatch = DeviceBatch.from_host(ctx, inputs, target, normalize="spektral")
        assert dbatch.n == hbatch.n and dbatch.n_graphs == hbatch.n_graphs
        assert np.array_equal(dbatch.x.numpy(), hbatch.x.numpy())
        assert np.array_equal(dbatch.a.rowptr.numpy(), hbatch.a.rowptr.numpy())
        assert np.array_equal(dbatch.a.colidx.numpy()[:dbatch.a.nnz], hbatch.a.colidx.numpy()[:hbatch.a.nnz])
        assert np.array_equal(dbatch.a.vals.numpy()[:dbatch.a.nnz], hbatch.a.vals.numpy()[:hbatch.a.nnz])
        assert np.array_equal(dbatch.y.numpy(), hbatch.y.numpy())
        assert np.array_equal(dbatch.seg.dev.numpy(), hbatch.seg.dev.numpy())
        assert np.array_equal(dbatch.seg.ids.numpy(), np.asarray(inputs[2], np.int32))      # the id vector i, written by the collate
        assert np.array_equal(hbatch.seg.ids.numpy(), np.asarray(inputs[2], np.int32))      # ... and rebuilt from graph_ptr
        l1 = m1.train_step(hbatch, None, lr=0.05)
        l2 = m2.train_step(dbatch, None, lr=0.05)
        assert l1 == l2
        n_batches += 1
    assert n_batches == 6
    for a, b in zip(m1.get_weights(), m2.get_weights()):
        assert np.array_equal(a, b)


def test_fit_loop_host_and_device_loaders_agree(ctx, capsys):
    """n4: gcnx.fit (the loop of gcn.py:364-385 with the per-step PiecewiseConstantDecay and the size-weighted
    per-epoch evaluation) gives the same history through the host loader and through the device-side loader."""
    import gcnx
    from gcnx import DisjointLoader, Graph, ListDataset, synth, DeviceDataset, DeviceDisjointLoader
    from gcnx.models import GCN2
    raw = synth.tiny_graphs(14, 16, seed=7)
    tr = ListDataset([Graph(x=x, a=a, y=y) for x, a, y in raw[:10]])
    te = ListDataset([Graph(x=x, a=a, y=y) for x, a, y in raw[10:]])
    epochs = 3
    runs = []
    for mode in ("host", "device"):
        m = GCN2(ctx, 2, hidden=16, use_graph=False, seed=2)
        if mode == "host":
            ltr = DisjointLoader(tr, batch_size=4, epochs=epochs, shuffle=True, seed=5)
            lte = DisjointLoader(te, batch_size=3, shuffle=False)
            out = gcnx.fit(m, ltr, lte, epochs=epochs, normalize="spektral")
        else:
            ltr = DeviceDisjointLoader(DeviceDataset(ctx, tr, normalize="spektral"), batch_size=4, epochs=epochs, shuffle=True, seed=5)
            lte = DeviceDisjointLoader(DeviceDataset(ctx, te, normalize="spektral"), batch_size=3, shuffle=False)
            out = gcnx.fit(m, ltr, lte, epochs=epochs)
        runs.append(out)
    assert "Ep. 3 - Loss:" in capsys.readouterr().out
    h0, h1 = np.array(runs[0]["history"]), np.array(runs[1]["history"])
    assert h0.shape == (epochs, 4) and np.array_equal(h0, h1)
    assert len(runs[0]["weights"]) == epochs and len(runs[0]["performance"]) == epochs
    for a, b in zip(runs[0]["weights"][-1], runs[1]["weights"][-1]):
        assert np.array_equal(a, b)


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


def test_stashed_metrics_equal_the_per_step_reads(ctx):
    """train_step(fetch="stash") + collect_metrics(): the (loss, accuracy) of every step without a host round trip per
    step -- same numbers as reading them after each step, across a ring wrap, eager and graph replay."""
    g = load_golden("gcn2_cfg1_tiny_weighted")
    for use_graph in (False, True):
        m, batch, hb = _model_from_golden(ctx, g, use_graph)
        ref = [m.train_step(batch, None, lr=0.02) for _ in range(7)]
        m2, batch2, _ = _model_from_golden(ctx, g, use_graph)
        m2._MRING = 4                                            # wraps after four steps
        for _ in range(7):
            assert m2.train_step(batch2, None, lr=0.02, fetch="stash") is None
        assert m2.collect_metrics() == ref and m2.collect_metrics() == []


def test_ecoli_config2_full_size_vs_cpu_restatement(ctx):
    """BASELINE config 2 (B=32, F=128, fp32) at full size against the fp32 C restatement."""
    from oracle import c_oracle
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2
    hb = synth.ecoli_shard(0, 32, 128, seed=1)            # the batch bench.py times at N = 1
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


def test_rccl_path_single_rank(ctx):
    """The RCCL leg of the C ABI (dlopen librccl, unique id, ncclCommInitRank, in-place fp32 all-reduce on the
    ctx stream) with one rank: sum and max of one rank are the identity.  Multi-rank runs need one GPU per rank
    (the driver's 8-GPU tier); the host sequence around this call is covered at world_size 2 by test_dist_cpu.py."""
    import ctypes as C
    from gcnx import _lib as L
    lib = ctx.lib
    uid = C.create_string_buffer(L.UNIQUE_ID_BYTES)
    L.check(lib.gcnx_comm_unique_id(uid))
    assert any(uid.raw)
    comm = C.c_void_p()
    L.check(lib.gcnx_comm_init_rank(ctx.h, uid.raw, 1, 0, C.byref(comm)), ctx.h)
    x = np.random.default_rng(0).standard_normal(33286).astype(np.float32)      # the GCN2(F=128) gradient buffer size
    d = ctx.to_device(x)
    for op in (L.RED_SUM, L.RED_MAX):
        L.check(lib.gcnx_allreduce_f32(ctx.h, comm, d.ptr, d.size, op), ctx.h)
        assert np.array_equal(d.numpy(), x)
    assert lib.gcnx_allreduce_f32(ctx.h, comm, d.ptr, d.size, 7) == 1            # bad op -> GCNX_ERR_INVALID
    # the collective recorded into a HIP graph (what a multi-GPU step does: gradients | ncclAllReduce | SGD, one launch)
    from gcnx import device as D
    p = ctx.to_device(np.ones(33286, np.float32))

    def seq():
        L.check(lib.gcnx_allreduce_f32(ctx.h, comm, d.ptr, d.size, L.RED_SUM), ctx.h)
        D.sgd(ctx, p, d, 0.5)
    graph = ctx.capture(seq)
    graph.launch(); graph.launch()
    assert np.allclose(p.numpy(), 1.0 - 2 * 0.5 * x, rtol=1e-6, atol=1e-6) and np.array_equal(d.numpy(), x)
    graph.destroy()
    lib.gcnx_comm_destroy(comm)
    # the Python wrapper with world_size 1 is a no-op communicator
    from gcnx.comm import Communicator
    c1 = Communicator(ctx, 0, 1)
    c1.allreduce_sum(d); c1.barrier()
    assert np.array_equal(d.numpy(), x) and float(c1.allreduce_host([3.5], "max")[0]) == 3.5


@pytest.mark.parametrize("hidden,f_in,prec", [(128, 128, "f32"), (256, 128, "f32"), (256, 256, "bf16")])
def test_gcn2_multi_gpu_step_graph_with_a_real_rccl_collective(ctx, hidden, f_in, prec):
    """The code path a rank of an N-GPU run takes -- gradients (fold-only reductions), ncclAllReduce and SGD recorded into
    ONE HIP graph -- with a real RCCL communicator.  One GPU is all there is here, so the communicator has one rank and
    the wrapper only CLAIMS a world of two: the collective is then the identity and the step must equal the plain
    single-process step bit for bit (same kernels, same order), through capture and replay.  hidden = 128: the one-launch
    layers with one flat all-reduce; hidden = 256: the two-launch layers, the all-reduce in two buckets -- the first on the
    side stream (a second branch of the captured graph) beside layer 1's backward; (256, 256, "bf16"): the same with the
    streaming bf16 GEMMs and bf16 storage of S1, Y1, dH2, dZ1 (r3) inside the captured, bucketed step."""
    import ctypes as C
    from gcnx import _lib as L, synth
    from gcnx.comm import Communicator
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2

    class OneRankPosingAsTwo(Communicator):
        def __init__(self, ctx):
            self.ctx, self.rank, self.world_size = ctx, 0, 2
            uid = C.create_string_buffer(L.UNIQUE_ID_BYTES)
            L.check(ctx.lib.gcnx_comm_unique_id(uid))
            h = C.c_void_p()
            ctx._ck(ctx.lib.gcnx_comm_init_rank(ctx.h, uid.raw, 1, 0, C.byref(h)))
            self.h, self._scratch, self._path = h, ctx.zeros(4), None

    # (hidden = 256: 130 graphs, so that the batch has a tile plan and takes the large-batch sequence the buckets live in)
    hb = synth.ecoli_batch(6 if hidden == 128 else 130, f_in, seed=21)
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)

    def run(comm):
        a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
        batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
        m = GCN2(ctx, 2, hidden=hidden, seed=5, comm=comm, prec=prec)
        out = [m.train_step(batch, None, lr=0.05, global_batch=hb.n_graphs) for _ in range(4)]   # eager, capture, replay x2
        assert bool(m._bufs.get("act16")) == (prec == "bf16")
        return m, out

    comm = OneRankPosingAsTwo(ctx)
    try:
        m2, o2 = run(comm)
        assert m2._comm_in_graph() and not getattr(m2, "_comm_capture_failed", False)
        assert (hidden == 128) == m2._fused(DeviceBatch(ctx, ctx.to_device(hb.x), DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr),
                                                        Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y)))
        assert m2._reduced_in_backward == (hidden == 256)
        m1, o1 = run(None)
        assert o1 == o2
        for w1, w2 in zip(m1.get_weights(), m2.get_weights()):
            assert np.array_equal(w1, w2)
    finally:
        comm.close()


@pytest.mark.parametrize("hidden,n_calls", [(32, 1), (48, 2)])
def test_gcn2_world_size_2_on_one_gpu_equals_single_rank(hidden, n_calls):
    """The sharded step on the DEVICE path at world_size 2: two ranks (threads, each with its own Context on device
    0, a host-mediated communicator of the Communicator interface, tests/thread_comm.py) take graph shards of one
    batch, normalise the loss by the global batch, all-reduce the flat gradient (+ loss/accuracy tail) and
    apply SGD -- and land on the 1-rank loss, accuracy, gradients and updated weights (fp32 reduction order).
    hidden = 32: the one-launch layers, ONE flat all-reduce; hidden = 48: the two-launch layers of large batches, where
    the all-reduce goes out in two buckets (r3) -- {dW2, db2, dW3, db3, metrics} as soon as layer 2's gradients are
    final, {dW1, db1} behind layer 1's backward."""
    import gcnx
    from gcnx import synth, shard
    from gcnx.models import DeviceBatch, GCN2
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from thread_comm import ThreadWorld
    hb = synth.ecoli_batch(6, 32, seed=8)
    hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)

    def make_batch(ctx, part):
        a = gcnx.DeviceCSR.from_host_csr(ctx, part.rowptr, part.colidx, part.vals, part.graph_ptr)
        seg = gcnx.Segments(ctx, part.graph_ptr)
        return DeviceBatch(ctx, ctx.to_device(part.x), a, seg, ctx.to_device(part.y, np.float32))

    ctx0 = gcnx.Context(0)
    ref = GCN2(ctx0, 2, hidden=hidden, seed=5, use_graph=False)
    ref_loss, ref_acc = ref.train_step(make_batch(ctx0, hb), None, lr=0.05)
    ref_g, ref_w = ref.gradients(), ref.get_weights()
    ctx0.close()

    def rank_fn(rank, make_comm):
        ctx = gcnx.Context(0)
        part, global_b = shard.shard_batch(hb, rank, 2)
        m = GCN2(ctx, 2, hidden=hidden, seed=5, use_graph=False, comm=make_comm(ctx))
        loss, acc = m.train_step(make_batch(ctx, part), None, lr=0.05, global_batch=global_b)
        out = (loss, acc, m.gradients(), m.get_weights(), m.comm.calls)
        ctx.close()
        return out

    res = ThreadWorld(2).run(rank_fn)
    for loss, acc, g, w, calls in res:
        assert calls == n_calls                            # one flat all-reduce, or the two buckets
        assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)) and acc == pytest.approx(ref_acc)
        for k in g:
            assert rel_err(g[k], ref_g[k]) < 2e-5, k
        for a, b in zip(w, ref_w):
            assert rel_err(a, b) < 2e-5
    for a, b in zip(res[0][3], res[1][3]):
        assert np.array_equal(a, b)                        # both ranks hold the same weights, bit for bit


def test_gcn2_world_size_2_large_batch_with_bf16_storage_equals_single_rank():
    """The same at the large-batch sequence with plain bf16 operands (r3): each rank's shard (~150 graphs, > 32 768 rows, a tile
    plan) stores S1, Y1, dH2, dZ1 as bfloat16 and reduces its gradients in two buckets; the two ranks' summed gradients
    equal the single-rank step on the whole batch up to fp32 reduction order -- the bf16 roundings are the same numbers
    on either side (per row: no rounding depends on which rank holds the row)."""
    import gcnx
    from gcnx import synth, shard
    from gcnx.models import DeviceBatch, GCN2
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from thread_comm import ThreadWorld
    hb = synth.ecoli_batch(300, 256, seed=9)          # (~150 graphs per rank: a tile plan needs 128)
    hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)

    def make_batch(ctx, part):
        a = gcnx.DeviceCSR.from_host_csr(ctx, part.rowptr, part.colidx, part.vals, part.graph_ptr)
        return DeviceBatch(ctx, ctx.to_device(part.x), a, gcnx.Segments(ctx, part.graph_ptr), ctx.to_device(part.y, np.float32))

    ctx0 = gcnx.Context(0)
    ref = GCN2(ctx0, 2, hidden=256, seed=5, use_graph=False, prec="bf16")
    ref_loss, ref_acc = ref.train_step(make_batch(ctx0, hb), None, lr=0.05)
    assert ref._bufs.get("act16")
    ref_g, ref_w = ref.gradients(), ref.get_weights()
    ctx0.close()

    def rank_fn(rank, make_comm):
        ctx = gcnx.Context(0)
        part, global_b = shard.shard_batch(hb, rank, 2)
        m = GCN2(ctx, 2, hidden=256, seed=5, use_graph=False, comm=make_comm(ctx), prec="bf16")
        loss, acc = m.train_step(make_batch(ctx, part), None, lr=0.05, global_batch=global_b)
        out = (loss, acc, m.gradients(), m.get_weights(), m.comm.calls, bool(m._bufs.get("act16")))
        ctx.close()
        return out

    res = ThreadWorld(2).run(rank_fn)
    for loss, acc, g, w, calls, act16 in res:
        assert calls == 2 and act16
        assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)) and acc == pytest.approx(ref_acc)
        for k in g:
            assert rel_err(g[k], ref_g[k]) < 5e-5, k
        for a, b in zip(w, ref_w):
            assert rel_err(a, b) < 5e-5
    for a, b in zip(res[0][3], res[1][3]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("aggregate,pool,connectivity", [("mean", "avg", "cat"), ("sum", "max", "cat"), ("mean", "sum", "cat"),
                                                         ("max", "sum", "cat"), ("min", "avg", "sum"), ("max", "max", "sum"),
                                                         ("prod", "sum", "cat")])
def test_general_gnn_aggregate_and_pool_options_match_oracle(ctx, aggregate, pool, connectivity):
    """GeneralGNN(aggregate="mean" | "max" | "min" | "prod", pool="avg" | "max") (Spektral options beside gcn.py:320's defaults; r3, "prod" r4): inference forward,
    training step (loss, probabilities, every gradient) against the fp64 oracle, which torch autograd pins for these options
    (tests/test_oracle.py).  Gradients at 1e-4 against the oracle on the device's side of every PReLU kink, as in the
    default-option test."""
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GeneralGNN
    from oracle import gcn_oracle as O
    hb, layers, flat = _general_gnn_case(31, 16, 32, 2, 5, connectivity=connectivity)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    m = GeneralGNN(ctx, 2, activation="softmax", hidden=32, message_passing=2, aggregate=aggregate, pool=pool,
                   connectivity=connectivity, use_graph=False)
    m.build(16)
    m.set_weights(flat, order="layer")
    x64, y64 = hb.x.astype(np.float64), hb.y.astype(np.float64)
    csr = (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None)
    rprobs, _, _ = O.general_gnn_forward(layers, x64, csr, hb.graph_ptr, False, aggregate=aggregate, pool=pool, connectivity=connectivity)
    assert rel_err(m(batch, training=False), rprobs) < TOL
    loss, acc = m.train_step(batch, None, lr=0.01)
    # (max / min aggregation and the max pool select an extremum: a selection has kinks of its own -- two candidates within
    # rounding of each other -- which the activation masks do not separate; 1.5e-4 measured on one tensor of "min": 3e-4 there)
    sel = aggregate in ("max", "min") or pool == "max"
    rl, ra, rp, _ = _assert_gnn_grads_kink_separated(m, m.gradients(), hb, layers, "f32", 3e-4 if sel else TOL,
                                                     f"GeneralGNN {aggregate}/{pool}/{connectivity}",
                                                     aggregate=aggregate, pool=pool, connectivity=connectivity)
    assert abs(loss - rl) < TOL * max(1, rl) and acc == pytest.approx(ra)
    assert rel_err(m._bufs["probs"].numpy(), rp) < TOL
    with pytest.raises(ValueError):
        GeneralGNN(ctx, 2, activation="softmax", aggregate="median")          # (not one of Spektral's five)


def test_general_gnn_sync_bn_world_size_2_equals_single_rank():
    """GeneralGNN with a communicator (sync-BN): two thread ranks with graph shards of one batch normalise with the
    GLOBAL batch statistics (all-reduced column sums in both moment passes and in the BN backward), and the step --
    loss, accuracy, every gradient, the updated weights and the moving statistics -- equals the single-rank step on
    the whole batch up to fp32 reduction order."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gcnx
    from gcnx import synth, shard
    from gcnx.models import DeviceBatch, GeneralGNN
    from thread_comm import ThreadWorld
    hb = synth.ecoli_batch(6, 16, seed=12)

    def make_batch(ctx, part):
        a = gcnx.DeviceCSR.from_host_csr(ctx, part.rowptr, part.colidx, None, part.graph_ptr)
        return DeviceBatch(ctx, ctx.to_device(part.x), a, gcnx.Segments(ctx, part.graph_ptr), ctx.to_device(part.y, np.float32))

    kw = dict(activation="softmax", hidden=32, message_passing=2, seed=3, use_graph=False)
    ctx0 = gcnx.Context(0)
    ref = GeneralGNN(ctx0, 2, **kw)
    ref_loss, ref_acc = ref.train_step(make_batch(ctx0, hb), None, lr=0.05)
    ref_g = ref.flat_g.numpy()[:ref.n_params]
    ref_w = [w.copy() for w in ref.get_weights()]
    ctx0.close()

    def rank_fn(rank, make_comm):
        ctx = gcnx.Context(0)
        part, global_b = shard.shard_batch(hb, rank, 2)
        m = GeneralGNN(ctx, 2, comm=make_comm(ctx), **kw)
        loss, acc = m.train_step(make_batch(ctx, part), None, lr=0.05, global_batch=global_b)
        out = (loss, acc, m.flat_g.numpy()[:m.n_params], [w.copy() for w in m.get_weights()])
        ctx.close()
        return out

    res = ThreadWorld(2).run(rank_fn)
    for loss, acc, g, w in res:
        assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)) and acc == pytest.approx(ref_acc)
        assert rel_err(g, ref_g) < 1e-4
        for a, b in zip(w, ref_w):
            assert rel_err(a, b) < 1e-4 or np.abs(a - b).max() < 1e-6
    for a, b in zip(res[0][3], res[1][3]):
        assert np.array_equal(a, b)                        # weights and moving statistics identical on both ranks


def test_evaluate_loop_matches_reference_semantics(ctx):
    """evaluate(loader) of gcn.py:342-362: eager forward per batch, loss/acc weighted by batch size."""
    from oracle import gcn_oracle as O
    from gcnx import DisjointLoader, Graph, ListDataset, synth
    from gcnx.models import GCN2, evaluate
    raw = synth.tiny_graphs(10, 16, seed=5)
    ds = ListDataset([Graph(x=x, a=a, y=y) for x, a, y in raw])
    loader = DisjointLoader(ds, batch_size=4, shuffle=False)            # epochs=None: infinite, like loader_te
    m = GCN2(ctx, 2, hidden=16, use_graph=False, seed=1)
    (loss, acc), preds = evaluate(m, loader, normalize="spektral")
    assert len(preds) == 3 and preds[0].shape == (4, 2) and preds[2].shape == (2, 2)
    w = dict(zip(ORDER, m.get_weights()))
    params = {k: v.astype(np.float64) for k, v in w.items()}
    tot_l = tot_a = 0.0
    for s0 in range(0, 10, 4):
        x, (idx, val, shape), i, y = O.disjoint_collate(raw[s0:s0 + 4])
        rp, ci = O.coo_to_csr(idx, shape[0])
        probs, _ = O.gcn2_forward(params, x.astype(np.float32).astype(np.float64), (rp, ci, O.gcn_filter_csr(rp, ci, None)),
                                  O.graph_ptr_from_ids(i, len(y)))
        tot_l += O.cce_loss(y.astype(np.float64), probs) * len(y); tot_a += O.categorical_accuracy(y, probs) * len(y)
    assert abs(loss - tot_l / 10) < TOL * max(1, tot_l / 10) and acc == pytest.approx(tot_a / 10)


def _general_gnn_case(seed, f_in, hidden, mp, n_graphs, tiny=False, **init_kw):
    from oracle import gcn_oracle as O
    from gcnx import synth
    rng = np.random.default_rng(seed)
    if tiny:
        graphs = synth.tiny_graphs(n_graphs, f_in, seed=seed, n_min=12, n_max=40)
        x, (idx, val, shape), i, y = O.disjoint_collate(graphs)
        rp, ci = O.coo_to_csr(idx, shape[0])
        hb = synth.HostBatch(x.astype(np.float32), rp.astype(np.int32), ci.astype(np.int32), None,
                             O.graph_ptr_from_ids(i, n_graphs).astype(np.int32), y.astype(np.float32))
    else:
        hb = synth.ecoli_batch(n_graphs, f_in, seed=seed)
    layers = O.general_gnn_init(rng, f_in, 2, hidden=hidden, message_passing=mp, pre=2, post=2, **init_kw)
    for grp in layers.values():
        for p in grp:                                     # move every parameter off its initial value
            for k in p:
                p[k] = p[k].astype(np.float32).astype(np.float64)
            if "alpha" in p:
                p["alpha"] = (0.25 * rng.random(p["alpha"].shape)).astype(np.float32).astype(np.float64)
            p["bias"] = (0.1 * rng.standard_normal(p["bias"].shape)).astype(np.float32).astype(np.float64)
            if "gamma" not in p:                          # batch_norm=False
                continue
            p["gamma"] = (1 + 0.1 * rng.standard_normal(p["gamma"].shape)).astype(np.float32).astype(np.float64)
            p["beta"] = (0.1 * rng.standard_normal(p["beta"].shape)).astype(np.float32).astype(np.float64)
            p["moving_mean"] = (0.1 * rng.standard_normal(p["moving_mean"].shape)).astype(np.float32).astype(np.float64)
            p["moving_var"] = (1 + 0.2 * rng.random(p["moving_var"].shape)).astype(np.float32).astype(np.float64)
    flat = [p[k] for g in ("pre", "gnn", "post") for p in layers[g]
            for k in ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_var", "alpha") if k in p]
    return hb, layers, flat


def _kink_free_case(f_in, hidden, mp, n_graphs, margin=5e-5, seeds=200):
    """A GeneralGNN case in which no BN output lies within `margin` of the PReLU kink, so that fp32 and fp64
    take the same branch everywhere and gradients are comparable at 1e-4 (seed scan on the host, deterministic)."""
    from oracle import gcn_oracle as O
    for seed in range(seeds):
        hb, layers, flat = _general_gnn_case(seed, f_in, hidden, mp, n_graphs, tiny=True)
        csr = (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None)
        _, caches, _ = O.general_gnn_forward(layers, hb.x.astype(np.float64), csr, hb.graph_ptr, True)
        zmin = min(np.abs(c["zb"]).min() for g in ("pre", "gnn") for c in caches[g])
        zmin = min(zmin, np.abs(caches["post"][0]["zb"]).min())
        if zmin > margin:
            return hb, layers, flat
    raise AssertionError("no kink-free case found")


KINK_REACH = {"f32": 2.0 ** -17, "bf16x3": 2.0 ** -11}     # how far from zero (in rms pre-activations) a flipped entry may lie


def _device_kink_masks(m, layers):
    """Which side of its PReLU / ReLU kink the DEVICE took every pre-activation to be on in the training step that just ran:
    zb = fma(z - mu, gamma * inv, beta) is the one expression both its forward and its backward kernels evaluate
    (csrc/bn.hip: bn_zb), from the Dense output z the model keeps per layer, the batch statistics (mu, inv) the step left and
    the layer's gamma / beta BEFORE the update -- so its sign is reproduced here exactly: the product (z - mu) * sc of two
    fp32 numbers is exact in fp64 and adding beta cannot change the sign of the exact sum.  None for a layer without a
    hidden activation."""
    masks = {"pre": [], "gnn": [], "post": []}
    i = 0
    for grp in ("pre", "gnn", "post"):
        for p in layers[grp]:
            L = m.layers[i]
            if L["act"] is None:
                masks[grp].append(None)
            else:
                z, mu, iv = m._bufs[f"z{i}"].numpy(), L["mean"].numpy(), L["inv"].numpy()
                ga = p["gamma"].astype(np.float32) if "gamma" in p else np.ones_like(iv)      # (batch_norm=False: identity transform)
                be = p["beta"].astype(np.float32) if "gamma" in p else np.zeros_like(iv)
                sc, d = ga * iv, z - mu                                                        # fp32, as on the device
                masks[grp].append(d.astype(np.float64) * sc.astype(np.float64) + be.astype(np.float64) > 0)
            i += 1
    return masks


def _assert_gnn_grads_kink_separated(m, got, hb, layers, prec, tol, what, drops=None, **okw):
    """Every gradient of the GeneralGNN step against the fp64 oracle evaluated on the DEVICE's side of every activation kink
    (oracle: dense_bn_act_bwd(pos=...)), at `tol` -- arithmetic error only; and the kink noise bounded separately: an entry
    the device and the fp64 reference put on different sides must lie within the precision's reach of zero
    (KINK_REACH x the layer's rms pre-activation), and such entries must be rare."""
    from oracle import gcn_oracle as O
    x64, y64 = hb.x.astype(np.float64), hb.y.astype(np.float64)
    csr = (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None)
    masks = _device_kink_masks(m, layers)
    _, caches, _ = O.general_gnn_forward(layers, x64, csr, hb.graph_ptr, True, drops=drops, **okw)
    nflip = ntot = 0
    for grp in ("pre", "gnn", "post"):
        for k, pos in enumerate(masks[grp]):
            if pos is None:
                continue
            zb = caches[grp][k]["zb"]                       # (with Dropout: already times the layer's factor; 0 where dropped)
            if drops is not None:
                pos = masks[grp][k] = pos & (drops[grp][k] > 0)
            flip = (zb > 0) != pos
            if flip.any():
                assert np.abs(zb[flip]).max() <= KINK_REACH[prec] * np.sqrt((zb ** 2).mean()), (what, grp, k, int(flip.sum()))
            nflip, ntot = nflip + int(flip.sum()), ntot + flip.size
    assert nflip <= max(4, 2e-4 * ntot), (what, nflip, ntot)
    okw = {k: v for k, v in okw.items() if k != "final_activation"}
    rl, ra, rg, rp, stats = O.general_gnn_loss_and_grads(layers, x64, csr, hb.graph_ptr, y64, drops=drops, masks=masks, **okw)
    li = 0
    for grp in ("pre", "gnn", "post"):
        for g in rg[grp]:
            assert set(got[li]) == set(g), (grp, li)
            # the Dense bias under BN has an analytically zero gradient (the device writes the exact zero, the oracle's sum
            # leaves fp64 noise): absolute slack relative to the layer's largest gradient
            layer_max = max(np.abs(v).max() for v in g.values())
            for name, ref in g.items():
                err = np.max(np.abs(got[li][name] - ref))
                assert err < tol * np.abs(ref).max() + 1e-6 * layer_max, (what, grp, li, name, err / max(np.abs(ref).max(), 1e-30))
                # (the report: against the tensor's own maximum -- or, for an analytically zero gradient, whose reference is fp64
                # summation noise, against the layer's largest gradient, which is what the assertion above grants it)
                zero_ref = np.abs(ref).max() < 1e-9 * layer_max
                PARITY_NOTES.append({"what": f"{what} {grp}{li} d{name}" + (" (analytically zero: error relative to the layer's largest gradient)" if zero_ref else ""),
                                     "rel_err": float(err / (layer_max if zero_ref else max(np.abs(ref).max(), 1e-30))),
                                     "tol": tol, "kink_flips": nflip})
            li += 1
    return rl, ra, rp, stats


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("f_in,hidden,mp,n_graphs,strict", [(16, 16, 4, 8, True), (16, 32, 2, 6, False), (16, 64, 4, 8, False),
                                                             (16, 256, 4, 3, True)])
def test_general_gnn_matches_oracle(ctx, f_in, hidden, mp, n_graphs, strict, prec):
    """The live model of gcn.py:320 (GeneralGNN: BN + PReLU + concat-skip + sum aggregation) on the
    device against the numpy oracle that test_oracle.py pins to torch autograd: training forward,
    loss, every gradient, moving statistics, SGD step, and the inference-mode forward.
    prec = "bf16x3" (r3): the Dense products on the split-bf16 panel kernels (csrc/gemm_panel.hip: weight images, the
    batch-norm moments out of the GEMM epilogue, dW as panels of the streaming kernel) at the same bars."""
    from oracle import gcn_oracle as O
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GeneralGNN
    if strict and hidden == 256:
        # the reference's own size (gcn.py:320: hidden=256, message_passing=4, NetSurfP width 16; concat width 1280,
        # 1.06 M parameters).  ~10^5 BN outputs: none within 2e-5 of the PReLU kink on the seed found (fp32 resolves
        # them at ~1e-6), so that the gradients can be held to the same bar as the forward pass.
        hb, layers, flat = _kink_free_case(f_in, hidden, mp, n_graphs, margin=2e-5, seeds=600)
    elif strict:
        hb, layers, flat = _kink_free_case(f_in, hidden, mp, n_graphs)
    else:
        hb, layers, flat = _general_gnn_case(3 + mp, f_in, hidden, mp, n_graphs)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    m = GeneralGNN(ctx, 2, activation="softmax", hidden=hidden, message_passing=mp, prec=prec)
    m.build(f_in)
    m.set_weights(flat, order="layer")
    assert all(np.array_equal(w, f.astype(np.float32)) for w, f in zip(m.get_weights(order="layer"), flat))
    csr = (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None)
    x64, y64 = hb.x.astype(np.float64), hb.y.astype(np.float64)
    # inference forward (moving statistics), evaluate() semantics of gcn.py:351
    rprobs, _, _ = O.general_gnn_forward(layers, x64, csr, hb.graph_ptr, False)
    assert rel_err(m(batch, training=False), rprobs) < TOL
    # training step.  Forward quantities are held to 1e-4.  Gradients (r4): a BN output that crosses the
    # PReLU kink between fp32 and fp64 moves a gradient by a whole term -- the fp32 run of the numpy oracle itself deviates
    # from its fp64 run by up to 2e-3 on the larger cases -- so every gradient is compared with the oracle's backward
    # evaluated on the DEVICE's side of every kink: 1e-4 (f32) / 2e-4 (bf16x3: 2^-18 per GEMM operand through ten
    # BatchNorm layers), on EVERY case, and the flipped entries are bounded by themselves (_assert_gnn_grads_kink_separated).
    before = m.get_weights(order="layer")
    loss, acc = m.train_step(batch, None, lr=0.01)
    got = m.gradients()
    rl, ra, rp, stats = _assert_gnn_grads_kink_separated(m, got, hb, layers, prec, TOL if prec == "f32" else 2e-4,
                                                        f"GeneralGNN {f_in}-{hidden}x{mp} B={n_graphs} {prec}")
    assert abs(loss - rl) < TOL * max(1, rl) and acc == pytest.approx(ra)
    assert rel_err(m._bufs["probs"].numpy(), rp) < TOL
    # moving statistics: m <- 0.99 m + 0.01 batch   (Keras momentum)
    after = m.get_weights(order="layer")
    it_b, it_a = iter(before), iter(after)
    si = 0
    for L in m.layers:
        for k in m.WEIGHT_ORDER:
            if k not in L:
                continue
            wb, wa = next(it_b), next(it_a)
            if k == "moving_mean":
                assert rel_err(wa, stats[si][0]) < TOL
            elif k == "moving_var":
                assert rel_err(wa, stats[si][1]) < TOL
                si += 1
            elif k == "kernel":
                assert np.allclose(wa, wb - np.float32(0.01) * got[m.layers.index(L)]["kernel"], rtol=0, atol=1e-6)


def test_gcn2_graph_policy_auto_is_eager_for_the_five_launch_step_and_equal_to_the_captured_one(ctx):
    """GCN2(use_graph="auto"), the default (r4): the five-launch step of E. coli-sized batches runs eagerly in one process (no ~10 us
    graph boundary per step), every other path replays a captured graph; eager and captured steps land on the same bits."""
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2
    hb = synth.ecoli_batch(6, 64, seed=2)
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    res = {}
    for opt in ("auto", True, False):
        m = GCN2(ctx, 2, hidden=64, seed=5, use_graph=opt)
        out = [m.train_step(batch, None, lr=0.02) for _ in range(4)]
        captured = sum(1 for g in m._graphs.values() if not isinstance(g, str))
        res[opt] = (out, [w.copy() for w in m.get_weights()], captured, m.use_graph)
    assert res["auto"][3] is False and res["auto"][2] == 0 and res[True][2] >= 1 and res[False][2] == 0
    for opt in (True, False):
        assert res[opt][0] == res["auto"][0]
        assert all(np.array_equal(u, v) for u, v in zip(res[opt][1], res["auto"][1]))
    m = GCN2(ctx, 2, hidden=144, seed=5)                    # (144 columns: not the one-launch layers -> captured)
    hb2 = synth.ecoli_batch(3, 144, seed=2)
    a2 = DeviceCSR.from_host_csr(ctx, hb2.rowptr, hb2.colidx, synth.gcn_norm_host(hb2.rowptr, hb2.colidx), hb2.graph_ptr)
    b2 = DeviceBatch(ctx, ctx.to_device(hb2.x), a2, Segments(ctx, hb2.graph_ptr), ctx.to_device(hb2.y))
    for _ in range(3):
        m.train_step(b2, None, lr=0.02)
    assert m.use_graph is True and sum(1 for g in m._graphs.values() if not isinstance(g, str)) >= 1


def test_general_gnn_with_spektrals_own_signature(ctx):
    """gcn.py:320 verbatim: `GeneralGNN(dataset.n_labels, activation="softmax")` -- no ctx: the process-wide default context --
    builds the same model as the explicit-ctx form and returns the same probabilities for the loader's host tuple."""
    import gcnx
    from gcnx import synth
    from gcnx.models import GeneralGNN
    hb = synth.ecoli_batch(3, 16, seed=5)
    m0 = GeneralGNN(2, activation="softmax", hidden=16, message_passing=2, seed=3)
    assert m0.ctx is gcnx.default_context() and m0.ctx is not ctx
    m1 = GeneralGNN(ctx, 2, activation="softmax", hidden=16, message_passing=2, seed=3)
    from gcnx.models import DeviceBatch
    from gcnx.device import DeviceCSR, Segments
    outs = []
    for m in (m0, m1):
        c = m.ctx
        a = DeviceCSR.from_host_csr(c, hb.rowptr, hb.colidx, None, hb.graph_ptr)
        b = DeviceBatch(c, c.to_device(hb.x), a, Segments(c, hb.graph_ptr), c.to_device(hb.y))
        outs.append(m(b, training=False))
    assert outs[0].shape == (3, 2) and np.array_equal(outs[0], outs[1])


def test_general_gnn_weight_orders(ctx):
    """get_weights() / set_weights() orders (model.get_weights(), gcn.py:383): "layer" lists every layer as Dense, BN,
    PReLU would; "keras" (default) differs for the four GeneralConv layers only -- one Keras Layer whose own kernel and
    bias come before its children's trainables (PReLU, then BatchNormalization), non-trainables last;
    "keras_children_first" is the alternative reading.  All three round-trip; PARITY UNPINNED (no Keras here)."""
    from gcnx.models import GeneralGNN
    m = GeneralGNN(ctx, 2, activation="softmax", hidden=8, message_passing=2, seed=1)
    m.build(4)
    rng = np.random.default_rng(0)
    for L in m.layers:                                    # make every array distinguishable
        for k in m.WEIGHT_ORDER:
            if k in L:
                L[k].copy_from_host(rng.standard_normal(L[k].shape).astype(np.float32))
    by_layer = m.get_weights(order="layer")
    assert [w.shape for w in by_layer[:7]] == [(4, 8), (8,), (8,), (8,), (8,), (8,), (8,)]
    keras = m.get_weights()
    n_pre = 2 * 7
    for a, b in zip(keras[:n_pre], by_layer[:n_pre]):      # the MLPs: identical in every order
        assert np.array_equal(a, b)
    g = by_layer[n_pre:n_pre + 7]                          # first GeneralConv: kernel,bias,gamma,beta,mm,mv,alpha
    want = [g[0], g[1], g[6], g[2], g[3], g[4], g[5]]      # kernel, bias | alpha | gamma, beta | mm, mv
    for a, b in zip(keras[n_pre:n_pre + 7], want):
        assert np.array_equal(a, b)
    cf = m.get_weights(order="keras_children_first")[n_pre:n_pre + 7]
    for a, b in zip(cf, [g[6], g[2], g[3], g[0], g[1], g[4], g[5]]):
        assert np.array_equal(a, b)
    for order in ("layer", "keras", "keras_children_first"):
        m2 = GeneralGNN(ctx, 2, activation="softmax", hidden=8, message_passing=2, seed=9)
        m2.build(4)
        m2.set_weights(m.get_weights(order=order), order=order)
        assert all(np.array_equal(a, b) for a, b in zip(m2.get_weights(order="layer"), by_layer))
    # the last post layer has no PReLU: 6 arrays
    assert len(by_layer) == 7 * 5 + 6 and len(keras) == len(by_layer)


def test_general_gnn_rejects_unbuilt_options(ctx):
    from gcnx.models import GeneralGNN
    with pytest.raises(ValueError):
        GeneralGNN(ctx, 2, activation="softmax", aggregate="median")
    with pytest.raises(NotImplementedError):
        GeneralGNN(ctx, 2, activation="sigmoid")
    with pytest.raises(NotImplementedError):
        GeneralGNN(ctx, 2, activation="softmax", connectivity="dense")
    with pytest.raises(NotImplementedError):
        GeneralGNN(ctx, 2, activation="softmax", hidden_activation="tanh")
    with pytest.raises(ValueError):
        GeneralGNN(ctx, 2, activation="softmax", dropout=1.0)


def test_general_gnn_linear_head_is_spektrals_default_activation(ctx):
    """GeneralGNN(output) with Spektral's default activation=None (r4; VERDICT r3 missing 3): model(inputs) returns the last
    post layer's BatchNormalization output -- the oracle's forward with final_activation=None -- in inference and after a
    training step; the step itself (from-logits cross-entropy, the form defined on a linear head) equals the softmax
    model's tf.function step: same loss, same gradients, bit for bit."""
    from oracle import gcn_oracle as O
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GeneralGNN
    hb, layers, flat = _general_gnn_case(9, 16, 32, 2, 5)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    csr = (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None)
    x64 = hb.x.astype(np.float64)
    lin = GeneralGNN(ctx, 2, hidden=32, message_passing=2, use_graph=False)             # activation=None, as Spektral defaults
    soft = GeneralGNN(ctx, 2, activation="softmax", hidden=32, message_passing=2, use_graph=False)
    for m in (lin, soft):
        m.build(16); m.set_weights(flat, order="layer")
    rlog, _, _ = O.general_gnn_forward(layers, x64, csr, hb.graph_ptr, False, final_activation=None)
    out = lin(batch, training=False)
    assert rel_err(out, rlog) < TOL and not np.allclose(out.sum(1), 1.0)                   # logits, not probabilities
    assert rel_err(soft(batch, training=False), O.softmax(rlog)) < TOL
    le, ae, oute = lin.evaluate_batch(batch, None)                 # (before any training step moves the moving statistics)
    assert rel_err(oute, rlog) < TOL
    assert abs(le - O.cce_loss_from_logits(hb.y.astype(np.float64), rlog)) < TOL * max(1.0, le)
    l1, a1 = lin.train_step(batch, None, lr=0.0)
    l2, a2 = soft.train_step(batch, None, lr=0.0)
    assert l1 == l2 and a1 == a2
    for ga, gb in zip(lin.gradients(), soft.gradients()):
        for k in ga:
            assert np.array_equal(ga[k], gb[k]), k


def _dropout_factors(ctx, model, rows_n, rows_b, step):
    """The Dropout factors (keep / (1 - rate)) the model's layers use at `step`, layer by layer, read back from the device
    generator itself (gcnx_dropout on a matrix of ones with the layer's stream id): {"pre": [...], "gnn": [...], "post": [...]}."""
    from gcnx import device as D
    out = {"pre": [], "gnn": [], "post": []}
    st = ctx.to_device(np.array([step], np.int32))
    for li, L in enumerate(model.layers):
        rows = rows_b if L["group"] == "post" else rows_n
        ones = ctx.to_device(np.ones((rows, L["fo"]), np.float32))
        D.dropout(ctx, ones, model.dropout, model.seed, li, st)
        out[L["group"]].append(ones.numpy().astype(np.float64))
    return out


@pytest.mark.parametrize("connectivity,batch_norm,act,rate,prec", [("sum", True, "prelu", 0.0, "f32"), ("cat", False, "prelu", 0.0, "f32"),
                                                                    ("cat", True, "relu", 0.0, "f32"), ("cat", True, "prelu", 0.3, "f32"),
                                                                    ("sum", False, None, 0.2, "f32"), ("sum", True, "relu", 0.5, "bf16x3")])
def test_general_gnn_connectivity_batch_norm_activation_dropout_options_match_oracle(ctx, connectivity, batch_norm, act, rate, prec):
    """GeneralGNN(connectivity="sum", batch_norm=False, hidden_activation="relu" | None, dropout > 0) -- Spektral options beside
    gcn.py:320's defaults (r3; VERDICT r2 missing 3): inference forward (Dropout inactive) and three training steps (eager, captured, replayed) (loss,
    probabilities, every gradient; the second step from the same captured-style sequence draws NEW masks) against the fp64
    oracle, which torch autograd pins for these options (tests/test_oracle.py) and which is fed the masks the device
    generator produced; the keep frequency is checked against 1 - rate.  hidden = 32 also runs the panel-GEMM path (bf16x3)."""
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GeneralGNN
    from oracle import gcn_oracle as O
    hid, mp = 32, 2
    hb, layers, flat = _general_gnn_case(41, 16, hid, mp, 5, connectivity=connectivity, batch_norm=batch_norm, hidden_activation=act)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    m = GeneralGNN(ctx, 2, activation="softmax", hidden=hid, message_passing=mp, connectivity=connectivity, batch_norm=batch_norm,
                   hidden_activation=act, dropout=rate, prec=prec, seed=7, use_graph=True)
    m.build(16)
    m.set_weights(flat, order="layer")
    assert m.n_params == sum(v.size for g in layers.values() for p in g for k, v in p.items() if not k.startswith("moving"))
    tol = TOL if prec == "f32" else 2e-4
    x64, y64 = hb.x.astype(np.float64), hb.y.astype(np.float64)
    csr = (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None)
    kw = dict(connectivity=connectivity, hidden_activation=act)
    rprobs, _, _ = O.general_gnn_forward(layers, x64, csr, hb.graph_ptr, False, **kw)
    assert rel_err(m(batch, training=False), rprobs) < tol
    for step in range(3):                                  # eager, captured, replayed
        drops = _dropout_factors(ctx, m, hb.n, hb.n_graphs, step) if rate > 0 else None
        if drops is not None:
            keep = np.concatenate([(d > 0).ravel() for g in drops.values() for d in g])
            assert abs(keep.mean() - (1 - rate)) < 0.02 and np.allclose(np.unique(np.concatenate([d.ravel() for g in drops.values() for d in g])),
                                                                        [0.0, 1 / (1 - rate)], rtol=1e-6)
            if step >= 1:
                assert not np.array_equal(drops["gnn"][0], prev["gnn"][0])          # a new mask every step
            prev = drops
        loss, acc = m.train_step(batch, None, lr=0.0)                                 # lr 0: the same weights in both steps
        rl, ra, rp, _ = _assert_gnn_grads_kink_separated(m, m.gradients(), hb, layers, prec, tol,
                                                         f"GeneralGNN {connectivity}/bn={batch_norm}/{act}/drop={rate} {prec} step {step}",
                                                         drops=drops, **kw)
        assert abs(loss - rl) < tol * max(1, rl) and acc == pytest.approx(ra), step
        assert rel_err(m._bufs["probs"].numpy(), rp) < tol




def test_directed_adjacency_takes_the_transposed_operator(ctx):
    """nothing in the reference's loader requires a symmetric adjacency (gcn.py:104-116 passes any scipy
    matrix).  A directed batch through the Spektral-shaped surface must be detected on the device (gcnx_csr_inspect)
    and the backward pass must aggregate with A^T, not A: gradients against the oracle's explicit transpose."""
    import scipy.sparse as sp
    from oracle import gcn_oracle as O
    from gcnx.loader import SparseTensor
    from gcnx.models import DeviceBatch
    rng = np.random.default_rng(21)
    sizes = [9, 14, 5]
    blocks = []
    for s in sizes:
        m = (rng.random((s, s)) < 0.3).astype(np.float64) * rng.uniform(0.2, 1.0, (s, s))   # directed, weighted
        np.fill_diagonal(m, 1.0)
        blocks.append(sp.csr_matrix(m))
    a = sp.block_diag(blocks).tocsr(); a.sort_indices()
    n = a.shape[0]
    coo = a.tocoo()
    order = np.lexsort((coo.col, coo.row))
    st = SparseTensor(np.stack([coo.row[order], coo.col[order]], 1).astype(np.int64), coo.data[order].astype(np.float32), (n, n))
    x = rng.standard_normal((n, 8)).astype(np.float32)
    i = np.repeat(np.arange(3), sizes)
    y = np.eye(2, dtype=np.float32)[[0, 1, 1]]
    batch = DeviceBatch.from_host(ctx, (x, st, i), y)
    assert batch.a.symmetric is False and batch.a.transpose() is not batch.a
    m = GCN2(ctx, 2, hidden=8, use_graph=False, seed=2)
    loss, acc = m.train_step(batch, None, lr=0.0)
    params = {k: v.astype(np.float64) for k, v in zip(ORDER, m.get_weights())}
    csr = (a.indptr.astype(np.int64), a.indices.astype(np.int64), st.values.astype(np.float64))
    rl, ra, rg, _ = O.gcn2_loss_and_grads(params, x.astype(np.float64), csr, np.concatenate([[0], np.cumsum(sizes)]),
                                          y.astype(np.float64))
    assert abs(loss - rl) < TOL * max(1, rl)
    for k, gk in m.gradients().items():
        assert rel_err(gk, rg[k]) < TOL, k
    # the symmetric shortcut on the same data would have been wrong (so the check is not vacuous)
    wrong = O.gcn2_loss_and_grads(params, x.astype(np.float64), csr, np.concatenate([[0], np.cumsum(sizes)]), y.astype(np.float64),
                                  csr_t=csr)[2]
    assert rel_err(wrong["w1"], rg["w1"]) > 1e-2
    # a symmetric batch is recognised as such
    sym = DeviceBatch.from_host(ctx, (x, sp.csr_matrix(a + a.T), i), y)
    assert sym.a.symmetric is True and sym.a.transpose() is sym.a


def test_adjacency_that_leaves_its_graph_block_is_refused(ctx):
    """the tile plan and the folded pool backward assume sp.block_diag structure; an entry outside its
    row's graph_ptr block (or a malformed graph_ptr) is a data error, reported at construction."""
    from gcnx import _lib as L
    from gcnx.device import DeviceCSR
    rowptr = np.array([0, 2, 4, 6, 8], np.int32)
    good = np.array([0, 1, 0, 1, 2, 3, 2, 3], np.int32)
    gp = np.array([0, 2, 4], np.int32)
    ok = DeviceCSR.from_host_csr(ctx, rowptr, good, None, gp)
    assert ok.inspect() == L.CSR_SYMMETRIC | L.CSR_BLOCK_DIAGONAL | L.CSR_GRAPH_PTR_OK
    bad = np.array([0, 2, 1, 3, 0, 2, 1, 3], np.int32)             # 0 <-> 2 and 1 <-> 3 cross the blocks (still symmetric)
    with pytest.raises(ValueError, match="not block-diagonal"):
        DeviceCSR.from_host_csr(ctx, rowptr, bad, None, gp)
    for bad_gp in ([1, 2, 4], [0, 2, 5], [0, 3, 2, 4]):
        with pytest.raises(ValueError, match="not block-diagonal"):
            DeviceCSR.from_host_csr(ctx, rowptr, good, None, np.array(bad_gp, np.int32))
    # without graph_ptr only symmetry is looked at; explicit symmetric=... skips the check altogether
    assert DeviceCSR.from_host_csr(ctx, rowptr, bad, None, None).symmetric is True
    assert DeviceCSR.from_host_csr(ctx, rowptr, bad, None, gp, symmetric=True).symmetric is True


def test_workspace_growth_does_not_invalidate_captured_graphs(ctx):
    """a captured step holds the ctx workspace pointer in its kernel arguments (split-K slabs, head
    partials).  An eager call that needs a larger workspace must not free that block under the graph: capture a step,
    force growth with a much larger problem, replay the graph -- same bits as an eager run of the same step."""
    from gcnx import device as D
    g = load_golden("gcn2_ecoli_mini_f16")
    m, batch, hb = _model_from_golden(ctx, g, use_graph=True)
    for _ in range(3):                                    # eager, capture + first replay, replay
        m.train_step(batch, None, lr=0.0)
    ref = m.gradients()
    # a split-K product whose slabs need far more workspace than anything the small step asked for
    rng = np.random.default_rng(0)
    xb = ctx.to_device(rng.standard_normal((300000, 256), dtype=np.float32))
    dwb = ctx.empty((256, 256))
    D.gemm_dw(ctx, xb, xb, dwb)
    D.gemm_dw(ctx, xb, xb, dwb, prec="bf16x3")
    ctx.sync()
    m.train_step(batch, None, lr=0.0)                      # replays the graph captured before the growth
    again = m.gradients()
    eager, ebatch, _ = _model_from_golden(ctx, g, use_graph=False)
    eager.train_step(ebatch, None, lr=0.0)
    for k in ORDER:
        assert np.array_equal(again[k], ref[k]) and np.array_equal(again[k], eager.gradients()[k]), k


# ---- BASELINE configs 3 and 5 as FULL train steps (VERDICT r1 "Next" item 1) ---------------------------------------
def _full_size_batch(workload):
    from gcnx import synth
    hb = synth.block_diag_batch() if workload == "block1m" else synth.power_law_batch()
    hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    return hb


def test_gcn2_bf16_storage_of_gemm_only_activations_changes_no_bit(ctx):
    """prec = "bf16" on a tile-plan batch (r3): S1 = A X, Y1, dH2 and dZ1 -- read by weight GEMMs only, which round their
    operands to bfloat16 anyway -- are stored as bfloat16 (gcnx_spmm_csr_bf16out, gcnx_gemm_fwd_bf16, ...).  Against the same
    model with fp32 storage (the knob GCNX_ACT16=0 sets): loss, accuracy, every gradient and the updated weights bit for
    bit, eagerly, captured and replayed; tall graphs (row chunks) and double-buffered tiles in the batch."""
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch
    hb = synth.block_diag_batch(100_000, 1_000_000, 256, seed=12)
    sizes = np.diff(hb.graph_ptr)
    assert len(sizes) >= 128 and (sizes > 1276).any() and (sizes <= 624).any()
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
    assert a.plan is not None
    y = np.zeros((len(sizes), 2), np.float32); y[np.arange(len(sizes)), np.arange(len(sizes)) % 2] = 1
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(y))
    res = {}
    for store16 in (True, False):
        m = GCN2(ctx, 2, hidden=256, seed=3, prec="bf16")
        m._knob["act16"] = store16
        out = []
        for step in range(3):                              # eager, captured, replayed
            loss, acc = m.train_step(batch, None, lr=0.01)
            out.append((loss, acc, {k: v.copy() for k, v in m.gradients().items()}, [w.copy() for w in m.get_weights()]))
        assert bool(m._bufs.get("act16")) == store16
        ev = m.evaluate_batch(batch, None)                 # the forward pass of evaluate() takes the same storage
        assert bool(m._bufs.get("act16")) == store16
        out.append((ev[0], ev[1], {"probs": ev[2]}, []))
        res[store16] = out
    for (l1, a1, g1, w1), (l0, a0, g0, w0) in zip(res[True], res[False]):
        assert l1 == l0 and a1 == a0
        for k in g1:
            assert np.array_equal(g1[k], g0[k]), k
        for u, v in zip(w1, w0):
            assert np.array_equal(u, v)
    assert np.isfinite(res[True][-1][0]) and any(np.abs(v).max() > 0 for v in res[True][0][2].values())


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_gcn2_pooled_layer_without_its_output_equals_the_two_launch_form(ctx, prec):
    """The pooled layer's forward leaves bits, pooled rows and counts and writes no Y2 on tile graphs (r3,
    gcnx_spmm_csr_relu_bits_pool + gcnx_pooled_dense_softmax_cce).  Against the same model with the aggregation, the pool
    and the head as before (the knob GCNX_POOL_IN_SPMM=0 sets): three steps -- eager, captured, replayed -- loss, accuracy,
    every gradient and the weights to summation order of the pool (1e-5)."""
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch
    hb = synth.block_diag_batch(100_000, 1_000_000, 256, seed=12)
    sizes = np.diff(hb.graph_ptr)
    assert len(sizes) >= 128 and (sizes > 1276).any() and (sizes <= 624).any()
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
    y = np.zeros((len(sizes), 2), np.float32); y[np.arange(len(sizes)), np.arange(len(sizes)) % 2] = 1
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(y))
    res = {}
    for fused in (True, False):
        m = GCN2(ctx, 2, hidden=256, seed=3, prec=prec, pool="avg" if prec == "f32" else "sum")
        m._knob["pool_in_spmm"] = fused
        out = []
        for step in range(3):
            loss, acc = m.train_step(batch, None, lr=0.01)
            out.append((loss, acc, {k: v.copy() for k, v in m.gradients().items()}, [w.copy() for w in m.get_weights()]))
        assert bool(m._bufs.get("pool_done")) == fused
        ev = m.evaluate_batch(batch, None)                 # the forward pass of evaluate(): the same launch, the head on its pooled rows
        assert bool(m._bufs.get("pool_done")) == fused
        out.append((ev[0], ev[1], {"probs": ev[2]}, []))
        res[fused] = out
    # (bf16 operands: a last-bit difference of a pooled sum -- the two forms add a graph's rows in different orders -- can move a
    # bf16 rounding of a later GEMM operand; 1.2e-5 measured on the weights after three steps with the r4 reduction tree)
    tol = 2e-5 if prec == "bf16" else 1e-5
    for (l1, a1, g1, w1), (l0, a0, g0, w0) in zip(res[True], res[False]):
        assert abs(l1 - l0) < tol * max(1.0, abs(l0)) and a1 == pytest.approx(a0)
        for k in g1:
            assert rel_err(g1[k], g0[k]) < tol, k
        for u, v in zip(w1, w0):
            assert rel_err(u, v) < tol


@pytest.mark.parametrize("workload", ["block1m", "powerlaw"])
def test_config3_and_config5_full_train_step_vs_c_oracle(ctx, workload):
    """BASELINE config 3 (1M nodes / 10M entries / F=256, 1 667 graphs) and config 5 (122 power-law graphs of 8 192
    nodes, max degree 4096, F=256) as whole train steps -- the tile-plan + side-stream path (config 3), hub rows
    split over waves (config 5), dW with K = 10^6 rows, the pool over 10^6 rows, the weighted aggregation at full size
    -- against the fp32 C restatement (oracle/gcn_oracle.c: blocked summation, so its own rounding stays far below
    the bar): loss, accuracy, every gradient and the SGD-updated weights against an INDEPENDENT implementation with its own
    ReLU masks -- a cross-check whose bounds (3e-4 f32, 6e-4 bf16x3 and plain bf16 against the oracle fed bf16-rounded GEMM
    operands) leave room for the kink noise of two evaluation orders; the 1e-4 bar itself is held by the kink-separated test.
    Plus a loose bound of plain bf16 against the fp32 oracle.  The first call runs eagerly, the second captures the step into a HIP graph, the third replays."""
    from oracle import c_oracle
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch
    hb = _full_size_batch(workload)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
    assert a.symmetric and a.plan is not None        # (config 5 too since r3: its plan lists the hub rows)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    m = GCN2(ctx, 2, hidden=256, seed=0)
    m.build(hb.f)
    w0 = m.get_weights()
    flat0 = np.concatenate([w.ravel() for w in w0])
    cpu = c_oracle.Gcn2Cpu(hb, 256, 2, flat0)
    ref = {}
    for key, bf in (("f32", False), ("bf16", True)):
        cpu.params[:] = flat0
        # (the bf16-operand MODEL follows the device's order of layer 1, (A X) W1 at F <= H: which operands get rounded)
        rl, ra = cpu.step(lr=0.0, bf16_operands=bf, layer1_s_order=bf)
        ref[key] = (rl, ra, cpu.grads.copy())
    lr = np.float32(0.05)
    # (r4) These bounds INCLUDE the ReLU-kink noise of comparing two evaluation orders with their own masks -- measured 0.9e-4
    # (f32) to 2.9e-4 (bf16x3, power-law) on dW1, of which the arithmetic error is 4e-7 / 3e-6: the 1e-4 bar of north_star is
    # held, for f32 AND bf16x3 and for both workloads, by the kink-separated test below, on the device's side of every kink.
    for prec, key, tol in (("f32", "f32", 3e-4), ("bf16x3", "f32", 6e-4), ("bf16", "bf16", 6e-4)):
        m.prec = prec
        m._drop_graphs()
        m.set_weights(w0)
        for _ in range(3):
            loss, acc = m.train_step(batch, None, lr=0.0)
        rl, ra, rg = ref[key]
        assert abs(loss - rl) < tol * max(1.0, abs(rl)), (prec, loss, rl)
        assert abs(acc - ra) <= 2.0 / hb.n_graphs, (prec, acc, ra)       # a graph on the decision boundary may flip
        got = np.concatenate([m.gradients()[k].ravel() for k in ORDER])
        off = 0
        for k, w in zip(ORDER, w0):                                       # per tensor, relative to its largest entry
            assert_close(got[off:off + w.size], rg[off:off + w.size], tol, f"{workload} full step {prec} d{k} vs C oracle ({key})")
            off += w.size
        if prec == "bf16":
            assert rel_err(got, ref["f32"][2]) < 3e-2                     # and bf16 stays close to the fp32 step
        # the update (gcn.py:338) on the same gradients
        m.train_step(batch, None, lr=float(lr))
        new = np.concatenate([w.ravel() for w in m.get_weights()])
        assert np.allclose(new, flat0 - lr * got, rtol=1e-6, atol=1e-7), prec
    # run-to-run determinism at full size (split-K slabs, two streams): same bits twice
    m.set_weights(w0); m.train_step(batch, None, lr=0.0)
    g1 = np.concatenate([m.gradients()[k].ravel() for k in ORDER])
    m.train_step(batch, None, lr=0.0)
    assert np.array_equal(g1, np.concatenate([m.gradients()[k].ravel() for k in ORDER]))


@pytest.mark.parametrize("workload", ["block1m", "powerlaw"])
def test_config3_full_step_vs_fp64_reference_with_the_relu_kinks_separated(ctx, workload):
    """(r4: also BASELINE config 5 -- 122 power-law graphs of 8 192 nodes, hub rows of up to 4096 entries -- whose bf16x3 step
    stood at 2.9e-4 against the C oracle's own masks: the same separation puts it at the 1e-4 bar.  Part (3), plain bf16
    operands, is config 3's precision and runs there only.)

    VERDICT r2, next 2.  What the 1.4e-4 on dW1 (bf16x3, full size, round 2) was: not arithmetic but ReLU KINKS.  The
    gradient of a ReLU network is discontinuous where a pre-activation is 0; an evaluation whose pre-activations differ
    from the reference's by delta lands on the other side for the ~N F rho(0) 2 delta entries within delta of zero, and
    each such flip moves dW1 = S1^T dZ1 by one whole term -- 1e-4 of max|dW1| here, because with random features dW1 is a
    heavily cancelling sum over 10^6 rows (E[dW1] = 0).  Measured (prec_diag3, r3): the fp32 path flips 21 of 5.1e8 masks
    (all |z| < 1e-7) and is 9.4e-5 from an fp64 reference on dW1 -- as is the fp32 C oracle (8.5e-5) -- and 4e-7 from the
    SAME reference evaluated with the device's masks; bf16x3 flips 635 (|z| < 6e-6): 1.6e-4 / 2.6e-6.

    So: fp64 reference (NumPy + scipy.sparse, the oracle's formulas) of the whole step, and every gradient compared on the
    SAME SIDE of every kink -- the reference backward evaluated with the device's masks [Y1 > 0], [Y2 > 0]:
      * f32 and bf16x3 at TOL = 1e-4 (north_star's bar; measured 5e-7 / 3e-6), the flipped entries counted and required to
        lie within the precision's reach of zero (|z_ref| <= 2^-20 / 2^-14 rms z);
      * against the reference's OWN masks both stay below 3e-4 (the kink noise itself; 0.9e-4 / 1.6e-4);
      * plain bf16 against an fp64 MODEL of bf16 operands that does NOT share the device's operand order -- every dense
        product of the reference sequence A (X W) with both operands rounded to bf16 (RNE), exact products and sums; the
        device computes layer 1 as (A X) W1 and rounds S1 = A X instead of X and A^T dZ1.  Both are the exact product
        plus independent relative roundings of rms u = 2^-9 / sqrt(3) per operand element, so two such evaluations of a
        gradient G = U^T V differ by 2 u sqrt(sum_r U^2 V^2) rms per entry: bound 6 sigma (65 536 entries) with the sum
        evaluated in fp64, relative to max|G| -- 2.6e-3 for dW1 (measured 3.7e-4), 1.2e-3 for dW2 (2.7e-5)."""
    import scipy.sparse as sp
    from oracle import gcn_oracle as O
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch
    hb = _full_size_batch(workload)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    m = GCN2(ctx, 2, hidden=256, seed=0, use_graph=False)
    m.build(hb.f)
    w0 = m.get_weights()
    A = sp.csr_matrix((hb.vals.astype(np.float64), hb.colidx, hb.rowptr), shape=(hb.n, hb.n))
    AT = A.T.tocsr()
    n, b = hb.n, hb.n_graphs
    P = sp.csr_matrix((np.ones(n), (np.repeat(np.arange(b), np.diff(hb.graph_ptr)), np.arange(n))), shape=(b, n))
    p = {k: w.astype(np.float64) for k, w in zip(ORDER, w0)}
    x, y = hb.x.astype(np.float64), hb.y.astype(np.float64)
    ident = lambda v: v

    def rb(v):                                    # RNE to bf16 and back (the oracle's bf16 model)
        return O.bf16_from_bits(O.bf16_bits(v.astype(np.float32))).astype(np.float64)

    def forward(r):                               # gcn.py:334 through GCNConv.call: A (X W) + b, relu
        z1 = A @ (r(x) @ r(p["w1"])) + p["b1"]; y1 = np.maximum(z1, 0)
        z2 = A @ (r(y1) @ r(p["w2"])) + p["b2"]; y2 = np.maximum(z2, 0)
        pooled = P @ y2
        return z1, y1, z2, pooled, pooled @ p["w3"] + p["b3"]

    def backward(fw, m1, m2, r):                  # SURVEY 8.A.6 with the masks given; from-logits CCE (tf.function form)
        z1, y1, z2, pooled, logits = fw
        loss, dl = O.cce(y, logits, O.softmax(logits), None, "logits")
        g = {"w3": pooled.T @ dl, "b3": dl.sum(0)}
        dz2 = (P.T @ (dl @ p["w3"].T)) * m2
        g["b2"] = dz2.sum(0); dh2 = AT @ dz2
        g["w2"] = r(y1).T @ r(dh2)
        dz1 = (r(dh2) @ r(p["w2"]).T) * m1
        g["b1"] = dz1.sum(0); dh1 = AT @ dz1
        g["w1"] = r(x).T @ r(dh1)
        return float(loss), g, (dh1, dh2)

    fw = forward(ident)
    own1, own2 = fw[0] > 0, fw[2] > 0
    loss64, g_own, _ = backward(fw, own1, own2, ident)
    rms1, rms2 = float(np.sqrt((fw[0] ** 2).mean())), float(np.sqrt((fw[2] ** 2).mean()))
    # (power-law batch: a hub row adds up to 4096 terms, and its pre-activation's rounding error grows with the row)
    reach = {"f32": 2.0 ** -20, "bf16x3": 2.0 ** -14} if workload == "block1m" else {"f32": 2.0 ** -18, "bf16x3": 2.0 ** -12}
    for prec in (("f32", "bf16x3", "bf16") if workload == "block1m" else ("f32", "bf16x3")):
        m.prec = prec; m._drop_graphs(); m.set_weights(w0)
        loss, _ = m.train_step(batch, None, lr=0.0)
        got = m.gradients()
        # (r3: with plain bf16 operands Y1 is stored as bfloat16; the pooled layer's launch leaves bits, pooled rows and counts,
        # Y2 rows are written for the graphs taller than a tile only: _device_relu_masks reads whichever form the step left)
        m1, m2 = _device_relu_masks(ctx, m, hb)
        if prec != "bf16":
            assert abs(loss - loss64) < TOL * abs(loss64), (prec, loss, loss64)
            f1, f2 = m1 != own1, m2 != own2
            # the masks differ only where the reference pre-activation is within the precision's reach of zero, and rarely
            assert int(f1.sum()) + int(f2.sum()) < (100 if prec == "f32" else 4000), (prec, int(f1.sum()), int(f2.sum()))
            assert not f1.any() or np.abs(fw[0][f1]).max() <= reach[prec] * rms1, prec
            assert not f2.any() or np.abs(fw[2][f2]).max() <= reach[prec] * rms2, prec
            _, g_dev, _ = backward(fw, m1, m2, ident)
            for k in ORDER:
                assert_close(got[k], g_dev[k], TOL, f"{workload} {prec} d{k} vs fp64 reference on the device's side of the ReLU kinks")
                assert rel_err(got[k], g_own[k]) < 3e-4, (prec, k)         # with the kink noise in: still there
        else:
            fwb = forward(rb)
            _, g_mod, (dh1, dh2) = backward(fwb, m1, m2, rb)
            u = 2.0 ** -9 / np.sqrt(3.0)
            for k, (uu, vv) in (("w1", (x, dh1)), ("w2", (fwb[1], dh2))):
                sig = 2.0 * u * np.sqrt(float(((uu * uu).T @ (vv * vv)).max()))
                bound = 6.0 * sig / float(np.abs(g_mod[k]).max())
                err = rel_err(got[k], g_mod[k])
                assert err < bound < 5e-3, (k, err, bound)
            for k in ("b1", "b2", "w3", "b3"):                             # no bf16 rounding of their own: propagated only
                assert rel_err(got[k], g_mod[k]) < 1e-4, k


@pytest.mark.parametrize("batch_norm,activation", [(True, "prelu"), (True, "relu"), (False, "relu"), (False, None), (False, "prelu")])
@pytest.mark.parametrize("aggregate", ["sum", "mean", "max", "prod"])
def test_general_conv_layer_surface(ctx, batch_norm, activation, aggregate):
    """spektral.layers.GeneralConv as a layer of its own (SURVEY 8(b) surface list; inside GeneralGNN at gcn.py:320):
    layer([x, a], training=) = sum-aggregation over a.indices of activation(BN(x W + b)) -- adjacency values ignored --
    and backward(dy), against the oracle's dense_bn_act + spmm; training updates the moving statistics, inference
    uses them."""
    from oracle import gcn_oracle as O
    from gcnx import synth
    from gcnx.device import DeviceCSR
    from gcnx.layers import GeneralConv
    hb = synth.ecoli_batch(2, 12, seed=5)
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)            # values present -- and ignored
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
    x = ctx.to_device(hb.x)
    conv = GeneralConv(20, batch_norm=batch_norm, activation=activation, aggregate=aggregate, seed=4)
    conv([x, a], training=True)                                  # builds
    minmax = aggregate in ("max", "min", "prod")
    agg = None if minmax else O.aggregate_vals(hb.rowptr, aggregate)   # None ("sum") / 1 / row length per entry ("mean", r3)
    rng = np.random.default_rng(1)
    names = list(conv.params) + list(conv.state)
    assert names == (["kernel", "bias"] + (["alpha"] if activation == "prelu" else []) +
                     (["gamma", "beta", "moving_mean", "moving_var"] if batch_norm else []))
    w = {k: (v.numpy() + 0.1 * rng.standard_normal(v.shape)).astype(np.float32) for k, v in {**conv.params, **conv.state}.items()}
    if "moving_var" in w:
        w["moving_var"] = np.abs(w["moving_var"]) + 0.5
    if "alpha" in w:
        w["alpha"] = (0.25 * rng.random(20)).astype(np.float32)
    conv.set_weights([w[k] for k in names])
    p = {k: v.astype(np.float64) for k, v in w.items()}
    if not batch_norm:
        p.update(gamma=np.ones(20), beta=np.zeros(20), moving_mean=np.zeros(20), moving_var=np.ones(20) - O.BN_EPS)
    x64 = hb.x.astype(np.float64)
    rp, ci = hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64)
    for training in ((False, True) if batch_norm else (False,)):
        y = conv([x, a], training=training)
        h, cache, mm, mv = O.dense_bn_act_fwd(x64, p, training, activation)
        if minmax:
            ry, rcnt = O.aggregate_prod(rp, ci, h) if aggregate == "prod" else O.aggregate_minmax(rp, ci, h, aggregate)
        else:
            ry = O.spmm_csr(rp, ci, agg, h)
        assert rel_err(y.numpy(), ry) < TOL, training
    dy = rng.standard_normal(y.shape).astype(np.float32)
    dx = conv.backward(ctx.to_device(dy))
    dh = ((O.aggregate_prod_bwd if aggregate == "prod" else O.aggregate_minmax_bwd)(rp, ci, h, ry, rcnt, dy.astype(np.float64)) if minmax else
          O.spmm_csr_T(rp, ci, agg, dy.astype(np.float64)))
    rdx, rg = O.dense_bn_act_bwd(dh, cache, p, activation)
    assert rel_err(dx.numpy(), rdx) < 2 * TOL
    for k in conv.grads:
        assert rel_err(conv.grads[k].numpy(), rg[k]) < 2 * TOL or np.abs(rg[k]).max() < 1e-9, k
    if batch_norm:
        assert rel_err(conv.state["moving_mean"].numpy(), mm) < TOL and rel_err(conv.state["moving_var"].numpy(), mv) < TOL
    with pytest.raises(ValueError):
        GeneralConv(8, aggregate="median")


def test_general_conv_layer_dropout(ctx):
    """GeneralConv(dropout=0.4): Dense -> BN -> Dropout -> PReLU -> aggregation in training mode, identity at inference; the
    layer's forward and backward against the oracle fed with the factors the device generator drew for that call
    (gcnx_dropout on ones: stream id = the layer's call count), a different mask on the next call."""
    from oracle import gcn_oracle as O
    from gcnx import synth, device as D
    from gcnx.device import DeviceCSR
    from gcnx.layers import GeneralConv
    hb = synth.ecoli_batch(2, 12, seed=6)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    x = ctx.to_device(hb.x)
    conv = GeneralConv(24, dropout=0.4, seed=9)
    y_inf = conv([x, a], training=False).numpy().copy()
    p = {k: v.numpy().astype(np.float64) for k, v in {**conv.params, **conv.state}.items()}
    x64 = hb.x.astype(np.float64)
    rp, ci = hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64)
    h, _, _, _ = O.dense_bn_act_fwd(x64, p, False, "prelu")
    assert rel_err(y_inf, O.spmm_csr(rp, ci, None, h)) < TOL                     # inference: no Dropout
    masks = []
    for call in (1, 2):
        ones = ctx.to_device(np.ones((hb.n, 24), np.float32))
        D.dropout(ctx, ones, 0.4, 9, call)
        drop = ones.numpy().astype(np.float64); masks.append(drop)
        assert abs((drop > 0).mean() - 0.6) < 0.05
        p = {k: v.numpy().astype(np.float64) for k, v in {**conv.params, **conv.state}.items()}      # (moving statistics move)
        y = conv([x, a], training=True)
        h, cache, _, _ = O.dense_bn_act_fwd(x64, p, True, "prelu", drop=drop)
        assert rel_err(y.numpy(), O.spmm_csr(rp, ci, None, h)) < TOL, call
        dy = np.random.default_rng(call).standard_normal(y.shape).astype(np.float32)
        dx = conv.backward(ctx.to_device(dy))
        rdx, rg = O.dense_bn_act_bwd(O.spmm_csr_T(rp, ci, None, dy.astype(np.float64)), cache, p, "prelu")
        assert rel_err(dx.numpy(), rdx) < 2 * TOL
        for k in conv.grads:
            assert rel_err(conv.grads[k].numpy(), rg[k]) < 2 * TOL or np.abs(rg[k]).max() < 1e-9, k
    assert not np.array_equal(masks[0], masks[1])


@pytest.mark.parametrize("model_kind", ["gcn2_fused", "gcn2_two_launch", "general_gnn"])
def test_learning_rate_from_a_device_scalar_serves_a_schedule_with_one_captured_step(ctx, model_kind):
    """gcnx_set_lr_source (r3; VERDICT r2 missing 5): with the rate read from a device scalar ONE captured step serves a
    schedule that changes every step; with the rate as a kernel argument every value captures its own graph.  Same weights
    bit for bit either way; and an eager gcnx_sgd of another user of the context still takes its argument afterwards."""
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GeneralGNN
    f = 16 if model_kind == "general_gnn" else 128
    hb = synth.ecoli_batch(4, f, seed=41)
    vals = None if model_kind == "general_gnn" else synth.gcn_norm_host(hb.rowptr, hb.colidx)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    rates = [0.02, 0.02, 0.015, 0.01, 0.0075, 0.005, 0.005, 0.0025]
    res = {}
    for on_device in (True, False):
        if model_kind == "general_gnn":
            m = GeneralGNN(ctx, 2, activation="softmax", hidden=32, message_passing=2, seed=3)
        else:
            m = GCN2(ctx, 2, hidden=128 if model_kind == "gcn2_fused" else 144, seed=3, use_graph=True)   # (the default, "auto", runs the five-launch step eagerly)
        m.lr_on_device = on_device
        out = [m.train_step(batch, None, lr=r) for r in rates]
        n_graphs = sum(1 for k, g in m._graphs.items() if k[0] == "grad" and not isinstance(g, str))
        res[on_device] = (out, [w.copy() for w in m.get_weights()], n_graphs)
    assert res[True][0] == res[False][0]
    for u, v in zip(res[True][1], res[False][1]):
        assert np.array_equal(u, v)
    assert res[True][2] == 1 and res[False][2] >= 2, (res[True][2], res[False][2])
    p = ctx.to_device(np.ones(8, np.float32)); g = ctx.to_device(np.ones(8, np.float32))
    D.sgd(ctx, p, g, 0.5)                                  # the context is back on by-value rates after a model's step
    assert np.array_equal(p.numpy(), np.full(8, 0.5, np.float32))


def test_general_gnn_captured_step_on_a_batch_with_a_tile_plan(ctx):
    """a batch of >= 128 graphs gets a gcnx_spmm_plan, whose per-rowptr row order must be bound OUTSIDE
    stream capture.  GeneralGNN asks for the unweighted view of the adjacency inside its step sequence; the view (and with
    it the binding) is now one object per operator, and binding is once per (plan, rowptr) -- so the second train_step,
    which captures, finds the order bound.  Three captured steps equal three eager steps; a GCN2 step captured on the
    same batch before keeps replaying correctly after the GeneralGNN has bound its own view (retired, not freed, arrays)."""
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2, GeneralGNN
    rng = np.random.default_rng(5)
    hb = synth.block_diag_batch(12000, 120000, 16, seed=7, mean_size=80)
    assert hb.n_graphs >= 128
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
    assert a.plan is not None and a.unweighted() is a.unweighted() and a.unweighted().plan == a.plan
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    g2 = GCN2(ctx, 2, hidden=64, use_graph=True)
    for _ in range(3):
        g2.train_step(batch, None, lr=0.0)
    ref2 = g2.gradients()
    runs = {}
    for use_graph in (False, True):
        m = GeneralGNN(ctx, 2, activation="softmax", hidden=64, message_passing=2, use_graph=use_graph, seed=3)
        out = [m.train_step(batch, None, lr=0.01) for _ in range(3)]
        runs[use_graph] = (out, m.get_weights())
    for (le, ae), (lg, ag) in zip(runs[False][0], runs[True][0]):
        assert le == lg and ae == ag
    for we, wg in zip(runs[False][1], runs[True][1]):
        assert np.array_equal(we, wg)
    # an explicit re-bind of the operator (row pointers "changed in place") while the GCN2 graph is alive: the graph still
    # replays into valid arrays
    a.rebind()
    g2.train_step(batch, None, lr=0.0)
    again = g2.gradients()
    for k in ref2:
        assert np.array_equal(again[k], ref2[k]), k


def _device_relu_masks(ctx, m, hb, hidden=256):
    """[Y1 > 0] and [Y2 > 0] of the GCN2 step that just ran, as the device holds them: Y1 as fp32 or (bf16 storage) bfloat16
    rows; Y2 rows for the graphs taller than a tile, the bit image -- word (slab, row), bit = column -- for the others."""
    from gcnx import device as _D
    b = m._bufs
    m1 = (_D.from_bf16(ctx, b["y1_16"]).numpy() if b.get("act16") else b["y1"].numpy()) > 0
    m2 = b["y2"].numpy() > 0
    if b.get("y2bits_ok"):
        img = b["y2bits"].numpy().view(np.uint32).reshape(hidden // 32, hb.n)
        from_bits = ((img.T[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool).reshape(hb.n, hidden)
        tile = np.repeat(np.diff(hb.graph_ptr) <= 1276, np.diff(hb.graph_ptr))
        m2 = np.where(tile[:, None], from_bits, m2)
    return m1, m2


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_config4_shards_of_the_1m_node_batch_sum_to_the_full_batch_step(ctx, prec):
    """BASELINE config 4 (SURVEY 8(e) acceptance) on the one GPU there is: the config-3 batch (1M nodes,
    10M entries, F = 256, 1 667 graphs) cut by gcnx.shard for world = 2, 4 and 8, every shard's step -- loss_and_grads with
    the loss normalised by the GLOBAL batch, exactly what a rank runs before its all-reduce -- executed on device 0 one after
    the other, the flat gradient buffers and the [loss, #correct] tail summed on the host in fp64 (what the RCCL all-reduce
    does in fp32), against the single-rank step on the whole batch: <= 2e-5 of each tensor's largest entry.  Every row of a
    shard is computed exactly as in the whole batch (rows sum in CSR order, a GEMM row does not depend on its tile), so the
    ReLU masks must be IDENTICAL -- asserted: no kink noise can hide in the bound.  fp32 and config 3's bf16 operands."""
    from gcnx import shard
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch
    hb = _full_size_batch("block1m")
    B = hb.n_graphs
    m = GCN2(ctx, 2, hidden=256, seed=0, prec=prec, use_graph=False)
    m.build(hb.f)
    w0 = m.get_weights()

    def step(part):
        a = DeviceCSR.from_host_csr(ctx, part.rowptr, part.colidx, part.vals, part.graph_ptr)
        batch = DeviceBatch(ctx, ctx.to_device(part.x), a, Segments(ctx, part.graph_ptr), ctx.to_device(part.y))
        m.loss_and_grads(batch, None, global_batch=B)
        flat = m.flat_g.numpy().astype(np.float64)                    # gradients + [loss sum / B, #correct]
        masks = _device_relu_masks(ctx, m, part)
        for arr in (batch.x, batch.y, a.rowptr, a.colidx, a.vals):
            arr.free()
        return flat, masks

    full, (f1, f2) = step(hb)
    assert f1.any() and f2.any()
    sizes = {k: int(np.prod(w.shape)) for k, w in zip(ORDER, w0)}
    for world in (2, 4, 8):
        bounds = shard.partition_graphs(hb.graph_ptr, hb.rowptr, world, hb.f)
        assert bounds[0] == 0 and bounds[-1] == B and np.all(np.diff(bounds) > 0)
        cost = (np.diff(hb.rowptr.astype(np.int64)[hb.graph_ptr[bounds]]) + np.diff(hb.graph_ptr[bounds].astype(np.int64)))
        assert cost.max() <= 1.02 * cost.mean(), (world, cost)         # cost-balanced contiguous ranges
        total = np.zeros_like(full)
        for r in range(world):
            part = hb.slice_graphs(int(bounds[r]), int(bounds[r + 1]))
            flat, (s1, s2) = step(part)
            r0, r1 = int(hb.graph_ptr[bounds[r]]), int(hb.graph_ptr[bounds[r + 1]])
            assert np.array_equal(s1, f1[r0:r1]) and np.array_equal(s2, f2[r0:r1]), (world, r)
            total += flat
        off = 0
        for k in ORDER:
            assert_close(total[off:off + sizes[k]], full[off:off + sizes[k]], 2e-5, f"config 4, {prec}, {world} shards summed: d{k}")
            off += sizes[k]
        assert abs(total[off] - full[off]) <= 2e-5 * abs(full[off]), (world, "loss")
        assert total[off + 1] == full[off + 1], (world, "#correct")
