"""CPU tests of the host side: Spektral-surface loader, sharding, workload generators, and the
C-ABI library (loads, exports every symbol include/gcnx.h declares, fails loudly without a GPU)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err
from oracle import gcn_oracle as O


def test_abi_exports_every_declared_symbol():
    from gcnx import _lib
    hdr = open(os.path.join(ROOT, "include", "gcnx.h")).read()
    declared = set(re.findall(r"GCNX_API\s+(?:const\s+char\s*\*|int64_t|int)\s+(gcnx_\w+)\s*\(", hdr))
    assert len(declared) >= 35
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (gcnx_\w+)", out))
    assert declared <= exported, declared - exported
    lib = _lib.load()                       # binds argtypes of every symbol
    assert lib.gcnx_version() >= 100


def test_abi_structs_match_their_ctypes_mirrors(tmp_path):
    """gcnx_pending_reduce and gcnx_head_args cross the boundary by pointer: the ctypes mirrors in gcnx/_lib.py must have
    the layout a C compiler gives the declarations of include/gcnx.h (size and every field offset)."""
    import ctypes as C
    from gcnx import _lib
    src = tmp_path / "layout.c"
    fields = {"gcnx_pending_reduce": [f for f, _ in _lib.PendingReduce._fields_],
              "gcnx_head_args": [f for f, _ in _lib.HeadArgs._fields_]}
    body = "".join(f'  printf("{st} %zu", sizeof({st}));\n' +
                   "".join(f'  printf(" %zu", offsetof({st}, {f}));\n' for f in fl) + '  printf("\\n");\n'
                   for st, fl in fields.items())
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gcnx.h"\nint main(void) {\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    for line, (st, cls) in zip(out, (("gcnx_pending_reduce", _lib.PendingReduce), ("gcnx_head_args", _lib.HeadArgs))):
        name, size, *offs = line.split()
        assert name == st and int(size) == C.sizeof(cls), (st, size, C.sizeof(cls))
        assert [int(o) for o in offs] == [getattr(cls, f).offset for f, _ in cls._fields_], st


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when no HIP device exists (this container)."""
    import ctypes as C
    from gcnx import _lib
    lib = _lib.load()
    n = C.c_int(-1)
    lib.gcnx_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    import gcnx
    with pytest.raises(_lib.GcnxError, match="no HIP device"):
        gcnx.Context(0)
    assert "no CPU fallback" in _lib.last_error()


def test_model_constructors_keep_spektrals_signature():
    """GeneralGNN(output, activation=None, hidden=256, message_passing=4, pre_process=2, post_process=2,
    connectivity="cat", batch_norm=True, dropout=0.0, aggregate="sum", hidden_activation="prelu", pool="sum") -- Spektral's names,
    order and defaults behind an OPTIONAL leading ctx (gcn.py:320 is `GeneralGNN(dataset.n_labels, activation="softmax")`; without
    a ctx the process-wide default context is used -- and without a GPU that raises: no CPU fallback)."""
    import inspect
    from gcnx.models import GCN2, GeneralGNN
    from gcnx.layers import GCNConv, GeneralConv
    ps = list(inspect.signature(GeneralGNN.__init__).parameters.values())
    assert [p.name for p in ps[:2]] == ["self", "ctx"]
    want = [("output", inspect.Parameter.empty), ("activation", None), ("hidden", 256), ("message_passing", 4), ("pre_process", 2),
            ("post_process", 2), ("connectivity", "cat"), ("batch_norm", True), ("dropout", 0.0), ("aggregate", "sum"),
            ("hidden_activation", "prelu"), ("pool", "sum")]
    assert [(p.name, p.default) for p in ps[2:2 + len(want)]] == want
    ps = list(inspect.signature(GeneralConv.__init__).parameters.values())[1:]
    assert [(p.name, p.default) for p in ps[:6]] == [("channels", 256), ("batch_norm", True), ("dropout", 0.0), ("aggregate", "sum"),
                                                      ("activation", "prelu"), ("use_bias", True)]
    ps = list(inspect.signature(GCNConv.__init__).parameters.values())[1:]
    assert [(p.name, p.default) for p in ps[:5]] == [("channels", inspect.Parameter.empty), ("activation", None), ("use_bias", True),
                                                      ("kernel_initializer", "glorot_uniform"), ("bias_initializer", "zeros")]
    if not os.path.exists("/dev/kfd"):
        for make in (lambda: GeneralGNN(2, activation="softmax"), lambda: GCN2(2)):
            with pytest.raises(RuntimeError):
                make()                                   # the default context needs a GPU


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gcn-string_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "gcn_oracle" not in src, f


def test_disjoint_loader_matches_spektral_semantics():
    from gcnx import DisjointLoader, Graph, ListDataset, synth
    raw = synth.tiny_graphs(7, 5, seed=1)
    ds = ListDataset([Graph(x=x, a=a, y=y) for x, a, y in raw])
    assert len(ds) == 7 and ds.n_labels == 2 and ds.n_node_features == 5
    loader = DisjointLoader(ds, batch_size=3, epochs=2, shuffle=False)
    assert loader.steps_per_epoch == 3
    batches = list(loader)
    assert len(batches) == 6                                   # 2 epochs x ceil(7/3)
    (x, a, i), y = batches[0]
    ox, (oidx, oval, oshape), oi, oy = O.disjoint_collate(raw[:3])
    assert np.array_equal(x, ox) and np.array_equal(a.indices, oidx) and np.array_equal(a.values, oval)
    assert a.dense_shape == oshape and np.array_equal(i, oi) and np.array_equal(y, oy)
    assert a.indices.dtype == np.int64 and i.dtype == np.int64 and x.dtype == np.float64
    assert batches[2][1].shape == (1, 2)                       # last batch of an epoch is smaller
    # epochs=None iterates forever; evaluate() counts steps (gcn.py:348-350)
    inf = DisjointLoader(ds, batch_size=4, shuffle=False)
    for _ in range(5):
        next(inf)
    # shuffle permutes graphs, boolean-mask indexing works like gcn.py:282-294
    mask = np.array([1, 0, 1, 1, 0, 0, 1], bool)
    assert len(ds[mask]) == 4 and len(ds[~mask]) == 3
    sh = DisjointLoader(ds, batch_size=7, epochs=1, shuffle=True, seed=0)
    (_, _, i2), _ = next(sh)
    assert len(i2) == len(np.concatenate([g[0] for g in raw]))
    sig = loader.tf_signature()
    assert sig[0][0][1] == (None, 5)


def test_gcnconv_preprocess_is_gcn_filter():
    from gcnx.layers import GCNConv
    from gcnx import synth
    x, a, y = synth.tiny_graphs(1, 4, seed=2)[0]
    got = GCNConv.preprocess(a)
    ref = O.gcn_filter_scipy(a, "spektral")
    assert rel_err(got.toarray(), ref.toarray()) < 1e-14
    assert np.isclose(got.toarray()[0, 0] * (a.sum(1)[0, 0] + 1), 2.0)   # diagonal 2/deg~


def test_synth_workloads_have_the_documented_shape():
    from gcnx import synth
    b = synth.ecoli_batch()
    assert b.n_graphs == 32 and b.f == 128 and 15000 < b.n < 26000 and 13 < b.nnz / b.n < 17
    rows = np.repeat(np.arange(b.n), np.diff(b.rowptr))
    gid = b.ids()
    assert np.all(gid[rows] == gid[b.colidx])                        # disjoint
    key = rows.astype(np.int64) * b.n + b.colidx
    assert np.all(np.diff(key) > 0)                                  # row-major sorted, unique
    tkey = b.colidx.astype(np.int64) * b.n + rows
    assert np.array_equal(np.sort(tkey), key)                        # symmetric
    assert np.all(np.isin(np.arange(b.n) * (b.n + 1), key))          # every self-loop stored
    c3 = synth.block_diag_batch(n=20000, nnz=200000, f=8, seed=2)
    assert c3.n == 20000 and c3.nnz == 200000 and abs(c3.n_graphs - 33) <= 1
    pl = synth.power_law_batch(n_graphs=1, graph_size=8192, f=4)
    assert np.diff(pl.rowptr).max() == 4096
    assert synth.spmm_algorithmic_bytes(1_000_000, 10_000_000, 256, False) == 4 * 1_000_001 + 40_000_000 + 2_048_000_000


def test_partition_balances_cost_and_covers_all_graphs():
    from gcnx import shard, synth
    b = synth.block_diag_batch(n=60000, nnz=600000, f=8, seed=5)
    for w in (1, 2, 3, 8):
        bounds = shard.partition_graphs(b.graph_ptr, b.rowptr, w, b.f)
        assert bounds[0] == 0 and bounds[-1] == b.n_graphs and np.all(np.diff(bounds) >= 1)
        costs = []
        for r in range(w):
            s = b.slice_graphs(int(bounds[r]), int(bounds[r + 1]))
            costs.append(s.nnz + s.n)
            assert s.rowptr[0] == 0 and s.rowptr[-1] == s.nnz and s.colidx.min() >= 0 and s.colidx.max() < s.n
        assert sum(costs) == b.nnz + b.n
        assert max(costs) / (sum(costs) / w) < 1.1
    # fewer graphs than ranks: trailing ranks get empty shards, nothing is lost
    small = synth.ecoli_batch(2, 8, seed=3)
    bounds = shard.partition_graphs(small.graph_ptr, small.rowptr, 4, 8)
    assert bounds[0] == 0 and bounds[-1] == 2 and np.all(np.diff(bounds) >= 0)


def test_sharded_oracle_step_equals_full_batch():
    """Host sharding logic end to end with the oracle as the compute."""
    from gcnx import shard, synth
    hb = synth.ecoli_batch(6, 8, seed=4)
    hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    rng = np.random.default_rng(0)
    params = O.gcn2_init(rng, 8, 8, 2)
    full = O.gcn2_loss_and_grads(params, hb.x.astype(np.float64), (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), hb.vals.astype(np.float64)),
                                 hb.graph_ptr, hb.y.astype(np.float64))
    tot = None; lsum = 0.0
    for r in range(3):
        s, gb = shard.shard_batch(hb, r, 3)
        l, _, g, _ = O.gcn2_loss_and_grads(params, s.x.astype(np.float64), (s.rowptr.astype(np.int64), s.colidx.astype(np.int64), s.vals.astype(np.float64)),
                                           s.graph_ptr, s.y.astype(np.float64), denom=gb)
        lsum += l
        tot = g if tot is None else {k: tot[k] + g[k] for k in g}
    assert abs(lsum - full[0]) < 1e-12
    for k in tot:
        assert rel_err(tot[k], full[2][k]) < 1e-12


def test_save_to_npz_round_trip_without_pickle(tmp_path):
    """gcn.py:59-64 mirror: same call, same keys; the per-epoch weight lists come back through plain arrays only."""
    import gcnx
    rng = np.random.default_rng(0)
    epochs = [[rng.standard_normal((4, 3)).astype(np.float32), rng.standard_normal(3).astype(np.float32)] for _ in range(3)]
    probas, labels = rng.random((10, 2)), np.eye(2)[rng.integers(0, 2, 10)]
    path = gcnx.save_to_npz(str(tmp_path), "run1", probas, labels, epochs, [0.5, 0.9, 0.7])
    with np.load(path, allow_pickle=False) as z:
        assert {"probas", "labels", "performance"} <= set(z.files)
        assert np.array_equal(z["probas"], probas) and np.array_equal(z["labels"], labels)
    assert gcnx.best_epoch(path) == 1
    for e in (0, 1, -1):
        got = gcnx.load_weights_npz(path, epoch=e)
        assert len(got) == 2 and all(np.array_equal(a, b) for a, b in zip(got, epochs[e]))


def test_piecewise_schedule_and_roc_auc_match_keras_and_sklearn():
    """n4 harness pieces: the per-step PiecewiseConstantDecay of gcn.py:321-324 (same values as the oracle's
    restatement) and roc_curve / auc against scikit-learn."""
    import gcnx
    from oracle import gcn_oracle as O
    sched = gcnx.PiecewiseConstantDecay.reference(20)
    assert [sched(s) for s in (0, 1, 6, 7, 100)] == [0.02, 0.002, 0.002, 0.0002, 0.0002]
    assert all(sched(s) == O.piecewise_lr(s, 20) for s in range(40))
    with pytest.raises(ValueError):
        gcnx.PiecewiseConstantDecay([0, 1], [0.1, 0.2])
    rng = np.random.default_rng(0)
    y = rng.integers(0, 2, 200)
    p = np.round(np.clip(0.3 * y + rng.random(200) * 0.8, 0, 1), 2)        # ties included
    fpr, tpr, thr = gcnx.roc_curve(y, p)
    from sklearn import metrics
    f2, t2, th2 = metrics.roc_curve(y, p, drop_intermediate=False)
    assert np.allclose(fpr, f2) and np.allclose(tpr, t2) and np.allclose(thr[1:], th2[1:])
    assert abs(gcnx.auc(fpr, tpr) - metrics.auc(f2, t2)) < 1e-12
    assert abs(gcnx.auc(fpr, tpr) - metrics.roc_auc_score(y, p)) < 1e-12


def test_bench_launcher_starts_ranks_itself_and_rendezvous_is_keyed_by_the_launcher(tmp_path):
    """`python bench.py --gpus N` invoked plainly must start the N rank processes itself (before
    anything touches the GPU), hand them a fresh run id for the RCCL rendezvous file, pass rank 0's JSON line through
    and fail if any rank fails.  CPU-side plumbing only (--selftest-launcher exchanges a fake id through the same
    rendezvous file the Communicator uses); world sizes 2 and 3."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GCNX_RUN_ID")}
    ids = []
    for world in (2, 3):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--selftest-launcher"],
                           capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0, r.stderr
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        assert rec == {"selftest": "launcher", "world": world, "ranks_seen": world, "ok": True, "run_id": rec["run_id"]}
        ids.append(rec["run_id"])
    assert ids[0] != ids[1] and len(ids[0]) == 32                # a fresh id per launch
    # a failing rank makes the launcher fail: WORLD_SIZE disagreeing with --gpus is refused by every rank
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"],
                         capture_output=True, text=True, env=dict(env, WORLD_SIZE="3", RANK="0"), timeout=60)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in bad.stderr
    # rendezvous key: the launcher's id when given, else torchrun's run id / restart count / port / agent pid
    from gcnx import comm
    p1 = comm._rendezvous_path({"GCNX_RUN_ID": "abc", "MASTER_PORT": "29500"})
    p2 = comm._rendezvous_path({"TORCHELASTIC_RUN_ID": "none", "MASTER_PORT": "29500"})
    p3 = comm._rendezvous_path({"TORCHELASTIC_RUN_ID": "none", "TORCHELASTIC_RESTART_COUNT": "1", "MASTER_PORT": "29500"})
    assert p1.endswith("gcnx_uid_abc_29500") and str(os.getppid()) not in os.path.basename(p1)
    assert p2 != p3 and os.path.basename(p2).endswith(f"_{os.getppid()}")
    # a launcher that exports neither (per-rank wrapper shells: no common parent): the ranks meet on the port alone ...
    p4 = comm._rendezvous_path({"MASTER_PORT": "29511"})
    assert os.path.basename(p4) == "gcnx_uid_port29511"
    # ... and a crashed earlier run's file under that key is never taken for this launch's id: rank 1 ignores a stale file
    # (older than the launch) and picks up the one rank 0 writes over it
    import threading
    import time
    path = str(tmp_path / "gcnx_uid_port29511")
    stale = b"S" * comm.L.UNIQUE_ID_BYTES
    with open(path, "wb") as fh:
        fh.write(stale)
    old = time.time() - 3600
    os.utime(path, (old, old))
    with pytest.raises(TimeoutError):
        comm.exchange_unique_id(1, 2, timeout_s=0.3, path=path)
    fresh = b"F" * comm.L.UNIQUE_ID_BYTES
    got = {}
    t = threading.Thread(target=lambda: got.setdefault("raw", comm.exchange_unique_id(1, 2, timeout_s=20, path=path)))
    t.start()
    time.sleep(0.2)
    assert comm.exchange_unique_id(0, 2, path=path, make_id=lambda: fresh) == fresh
    t.join(30)
    assert got.get("raw") == fresh


def test_bench_n_gpu_line_carries_the_config4_strong_scaling_key():
    """`bench.py --gpus N` (N > 1) reports BASELINE config 4 -- the 1M-node batch strong-scaled over the
    ranks -- as `config4_step` next to the weak-scaled headline.  CPU-side check through the real launcher: the key is emitted
    and the cut it describes covers the batch in cost-balanced contiguous ranges (timings need GPUs: None here)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GCNX_RUN_ID")}
    for world in (2, 8):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--selftest-config4"],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        recs = [json.loads(l) for l in r.stdout.strip().splitlines()]
        c4 = [x for x in recs if x.get("selftest") == "config4"][0]["config4_step"]
        assert c4["scaling"] == "strong" and c4["n_gpus"] == world and len(c4["shards"]) == world
        assert sum(s["graphs"] for s in c4["shards"]) == c4["global_graphs"]
        assert sum(s["nodes"] for s in c4["shards"]) == 1_000_000 and sum(s["entries"] for s in c4["shards"]) == 10_000_000
        cost = np.array([s["nodes"] + s["entries"] for s in c4["shards"]], np.float64)
        assert cost.max() <= 1.02 * cost.mean()
        assert [x for x in recs if x.get("selftest") == "launcher"][0]["ok"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'rec["config4_step"] = cfg4' in src and "config4_extra(ctx, comm, rank, world" in src


def test_shardable_generators_do_not_depend_on_the_number_of_ranks():
    """bench.py: every rank builds only its own graphs; graph g comes from its own random stream, so the global
    batch is the same for every world size, and the per-graph sizes (all a partition needs) are available without
    building any graph."""
    from gcnx import shard, synth
    sizes = synth.ecoli_sizes(12, seed=1)
    full = synth.ecoli_shard(0, 12, 8, seed=1)
    assert np.array_equal(np.diff(full.graph_ptr), sizes)
    bounds = shard.partition_by_cost(sizes * 16.0, 3)
    off = 0
    for r in range(3):
        part = synth.ecoli_shard(int(bounds[r]), int(bounds[r + 1]), 8, seed=1)
        ref = full.slice_graphs(int(bounds[r]), int(bounds[r + 1]))
        for k in ("x", "rowptr", "colidx", "graph_ptr", "y"):
            assert np.array_equal(getattr(part, k), getattr(ref, k)), k
        off += part.n
    assert off == full.n
    sz, pairs = synth.block_diag_plan(20000, 200000, seed=2)
    assert sz.sum() == 20000 and sz.sum() + 2 * pairs.sum() == 200000
    whole = synth.block_diag_shard(0, len(sz), sz, pairs, 4, seed=2)
    assert whole.n == 20000 and whole.nnz == 200000
    mid = synth.block_diag_shard(5, 9, sz, pairs, 4, seed=2)
    ref = whole.slice_graphs(5, 9)
    for k in ("x", "rowptr", "colidx", "graph_ptr", "y"):
        assert np.array_equal(getattr(mid, k), getattr(ref, k)), k
    a = (whole.rowptr, whole.colidx)
    rows = np.repeat(np.arange(whole.n), np.diff(whole.rowptr))
    g_of = np.repeat(np.arange(len(sz)), sz)
    assert np.array_equal(g_of[rows], g_of[whole.colidx])          # block-diagonal
    pl = synth.power_law_batch(3, 512, 4, seed=3, max_deg=128, first_graph=0)
    assert 128 <= np.diff(pl.rowptr).max() <= 132 and np.array_equal(       # the wired hub row (+ a stray Chung-Lu partner)
        synth.power_law_batch(2, 512, 4, seed=3, max_deg=128, first_graph=1).x, pl.x[512:])


def test_from_networkx_reproduces_the_reference_data_contract():
    """VERDICT r1 item 9 / SURVEY 8(a) a8: MyDataset's per-graph steps (gcn.py:104-128, 161-197) on networkx graphs
    built here (no pickle anywhere): node labels -> integers in node order, ``weight`` stripped so the adjacency is
    0/1, the builder's self-loops kept, scipy CSR int64, features stacked in node order as float64, one-hot label,
    edge attributes dropped unless requested -- and the graphs feed DisjointLoader like any other."""
    import networkx as nx
    import scipy.sparse as sp
    from gcnx import DisjointLoader, NetworkxDataset, from_networkx
    rng = np.random.default_rng(0)
    graphs, labels = [], []
    for k, n in enumerate((5, 8, 3)):
        g = nx.Graph()
        names = [f"res{n - i}" for i in range(n)]                       # string names, NOT in sorted order
        for i, nm in enumerate(names):
            g.add_node(nm, x=rng.standard_normal(16))
            g.add_edge(nm, nm, weight=0.0, dca=0.0, proximity=1.0)     # the 0-Angstrom self contact (gcn_utills.py:224-227)
        for i in range(n - 1):
            g.add_edge(names[i], names[i + 1], weight=3.8 + i, dca=0.1 * i, proximity=0.5)
        g.add_edge(names[0], names[-1], weight=9.5, dca=0.9, proximity=0.1)
        graphs.append(g); labels.append([1, 0] if k % 2 else [0, 1])
    sg = from_networkx(graphs[1], labels[1])
    n = 8
    assert sp.issparse(sg.a) and sg.a.dtype == np.int64 and sg.a.shape == (n, n)
    dense = sg.a.toarray()
    assert set(np.unique(dense)) == {0, 1}                              # weights stripped: a 0/1 pattern
    assert np.array_equal(dense, dense.T) and np.all(np.diag(dense) == 1)   # undirected, self-loops kept
    assert dense.sum() == n + 2 * (n - 1) + 2 - (2 if n == 2 else 0)
    assert dense[0, 1] == 1 and dense[0, n - 1] == 1 and dense[0, 2] == 0   # node ORDER of the graph, not of the names
    assert sg.x.dtype == np.float64 and sg.x.shape == (n, 16)
    assert np.array_equal(sg.x[0], graphs[1].nodes["res8"]["x"]) and np.array_equal(sg.x[-1], graphs[1].nodes["res1"]["x"])
    assert np.array_equal(sg.y, [1, 0]) and sg.e is None
    assert "weight" in graphs[1].edges["res8", "res7"]                  # the caller's graph is left untouched
    se = from_networkx(graphs[1], labels[1], use_edge_data=True)
    assert se.e.shape == (graphs[1].number_of_edges(), 2)              # dca, proximity (weight removed)
    ds = NetworkxDataset(graphs, labels, n_samples=3)
    assert len(ds) == 3 and ds.n_labels == 2 and ds.n_node_features == 16
    (x, a, i), y = next(DisjointLoader(ds, batch_size=3, epochs=1, shuffle=False))
    assert x.shape == (16, 16) and a.dense_shape == (16, 16) and np.array_equal(np.bincount(i), [5, 8, 3])
    assert a.indices.dtype == np.int64 and np.all(a.values == 1) and y.shape == (3, 2)
    # edge features requested (gcn.py:173-180 with use_edge_data): the loader yields Spektral's (x, a, e, i), e = the graphs'
    # [n_edges, S] arrays stacked in graph order, and tf_signature() describes it (r4; VERDICT r3 missing 4)
    dse = NetworkxDataset(graphs, labels, n_samples=3, use_edge_data=True)
    lde = DisjointLoader(dse, batch_size=3, epochs=1, shuffle=False)
    (xe, ae, ee, ie), ye = next(lde)
    assert np.array_equal(xe, x) and np.array_equal(ie, i) and np.array_equal(ae.indices, a.indices) and np.array_equal(ye, y)
    assert ee.shape == (sum(g.number_of_edges() for g in graphs), 2)
    assert np.array_equal(ee[:graphs[0].number_of_edges()], from_networkx(graphs[0], labels[0], use_edge_data=True).e)
    assert [t[0] for t in lde.tf_signature()[0]] == ["x", "a", "e", "i"] and lde.tf_signature()[0][2][1] == (None, 2)
    assert [t[0] for t in DisjointLoader(ds, batch_size=3).tf_signature()[0]] == ["x", "a", "i"]
    # a directed graph gives an asymmetric adjacency: allowed by the contract (the device side then transposes)
    dg = nx.DiGraph(); dg.add_node(0, x=np.zeros(2)); dg.add_node(1, x=np.ones(2)); dg.add_edge(0, 1, weight=2.0)
    assert np.array_equal(from_networkx(dg, [1, 0]).a.toarray(), [[0, 1], [0, 0]])


def test_collate_builds_the_disjoint_coo_directly_and_equals_block_diag_find_reorder():
    """DisjointLoader.collate (gcn.py:316-317, 350, 367) builds the row-major COO of the disjoint union by concatenating the
    per-graph CSR triples (r3: 42 -> 3 ms for a 32-graph E. coli batch); it must equal the Spektral-order restatement --
    to_disjoint (vstack, block_diag, repeat) + sp_matrix_to_sp_tensor (find, reorder) -- exactly: unsorted column indices,
    explicitly stored zeros (sp.find drops them), a COO input, an empty graph-free batch edge, single-node graphs."""
    import scipy.sparse as sp
    from gcnx.loader import Graph, collate_disjoint, to_disjoint, sp_matrix_to_sp_tensor, _disjoint_coo
    rng = np.random.default_rng(3)
    graphs = []
    for k, n in enumerate([1, 7, 30, 2, 64, 5]):
        d = (rng.random((n, n)) < 0.3).astype(np.float64)
        d = np.maximum(d, d.T); np.fill_diagonal(d, 1.0)
        a = sp.csr_matrix(d)
        if k == 2:                                   # unsorted indices inside the rows
            perm = a.copy(); perm.indices = perm.indices.copy()
            for r in range(n):
                lo, hi = perm.indptr[r], perm.indptr[r + 1]
                perm.indices[lo:hi] = perm.indices[lo:hi][::-1]; perm.data[lo:hi] = perm.data[lo:hi][::-1]
            perm.has_sorted_indices = False
            a = perm
        if k == 3:
            a = a.tocoo()
        if k == 4:                                   # explicit zeros stay in the structure of a CSR until someone prunes them
            a.data[::5] = 0.0
        graphs.append(Graph(x=rng.standard_normal((n, 4)), a=a, y=np.eye(2)[k % 2]))
    (x, st, i), y = collate_disjoint(graphs)
    x2, a2, i2 = to_disjoint([g.x for g in graphs], [g.a for g in graphs])
    st2 = sp_matrix_to_sp_tensor(a2)
    assert np.array_equal(x, x2) and np.array_equal(i, i2) and st.dense_shape == st2.dense_shape
    assert st.indices.dtype == np.int64 and np.array_equal(st.indices, st2.indices) and np.array_equal(st.values, st2.values)
    assert y.shape == (6, 2)
    e = _disjoint_coo([])
    assert e.indices.shape == (0, 2) and e.dense_shape == (0, 0)
