"""GPU parity tests proper: every hot-path entry point of the C ABI (include/gcnx.h), called
through ctypes, against the CPU oracle on the same seeded inputs.  Floating point, so the bar is
north_star's 1e-4 relative (max|diff|/max|ref|); most kernels are held to 1e-5."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4      # north_star: outputs within 1e-4 relative fp32 tolerance
TIGHT = 2e-5


def O():
    from oracle import gcn_oracle
    return gcn_oracle


def _csr(ctx, hb, weighted):
    from gcnx import synth
    from gcnx.device import DeviceCSR
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx) if weighted else None
    return DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr), vals


def _ref_spmm(hb, vals, h, bias=None, relu=False):
    o = O()
    out = o.spmm_csr(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64),
                     None if vals is None else vals.astype(np.float64), h.astype(np.float64))
    if bias is not None:
        out = out + bias.astype(np.float64)
    return np.maximum(out, 0) if relu else out


@pytest.mark.parametrize("f", [4, 10, 16, 32, 64, 128, 256, 512])
@pytest.mark.parametrize("weighted", [False, True])
def test_spmm_parity_widths(ctx, f, weighted):
    from gcnx import device as D, synth
    hb = synth.ecoli_batch(3, f, seed=f)
    a, vals = _csr(ctx, hb, weighted)
    rng = np.random.default_rng(f)
    bias = rng.standard_normal(f).astype(np.float32)
    h = ctx.to_device(hb.x)
    out = ctx.empty((hb.n, f))
    D.spmm(ctx, a, h, ctx.to_device(bias), out, act="relu")
    assert rel_err(out.numpy(), _ref_spmm(hb, vals, hb.x, bias, True)) < TIGHT
    D.spmm(ctx, a, h, None, out, act=None)
    assert rel_err(out.numpy(), _ref_spmm(hb, vals, hb.x)) < TIGHT


def test_spmm_long_rows_overflowing_the_lds_stage(ctx):
    """config 5 shape: one 4096-entry row next to short rows (chunk > LDS staging capacity)."""
    from gcnx import device as D, synth
    hb = synth.power_law_batch(n_graphs=1, graph_size=8192, f=64, seed=3)
    assert np.diff(hb.rowptr).max() == 4096
    for weighted in (False, True):
        a, vals = _csr(ctx, hb, weighted)
        out = ctx.empty((hb.n, 64))
        D.spmm(ctx, a, ctx.to_device(hb.x), None, out)
        assert rel_err(out.numpy(), _ref_spmm(hb, vals, hb.x)) < TIGHT


@pytest.mark.parametrize("weighted", [False, True])
def test_spmm_tile_kernel_all_tiers_forced(ctx, weighted, monkeypatch):
    """The LDS-tile kernels on a batch that holds every scheduling class: graphs at and around the limits of the
    pipelined kernel (624 rows: 32-column slabs; 1024: 16-column slabs; taller: row chunks) and of the round-1
    tier kernels (632 / 1276 since r3; 604 / 1236 before), taller ones, a single-node graph, rows with > 16 and > 32 entries (second register set / tail from global memory;
    on-demand index fetch) -- pipelined kernel, tier kernels and row gather against the oracle."""
    from gcnx import device as D, synth
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    sizes = [1, 40, 100, 604, 605, 624, 625, 632, 633, 768, 769, 1024, 1025, 1236, 1237, 1264, 1265, 1276, 1277, 300, 2000, 7]
    blocks = []
    for i, m in enumerate(sizes):
        dens = 0.5 if m in (40, 100) else min(1.0, 9.0 / m)   # ~20-entry rows (40 nodes) and ~50-entry rows (100 nodes)
        a = sp.random(m, m, density=dens, random_state=i, format="csr")
        a = ((a + a.T) > 0).astype(np.float32) + sp.identity(m, dtype=np.float32, format="csr")
        blocks.append((a > 0).astype(np.float32))
    a = sp.block_diag(blocks).tocsr(); a.sort_indices()
    n = a.shape[0]
    gp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    hb = synth.HostBatch(rng.standard_normal((n, 64), dtype=np.float32), a.indptr.astype(np.int32), a.indices.astype(np.int32),
                         None, gp, np.zeros((len(sizes), 2), np.float32))
    csr, vals = _csr(ctx, hb, weighted)
    bias = rng.standard_normal(64).astype(np.float32)
    ref = _ref_spmm(hb, vals, hb.x, bias, True)
    out = ctx.zeros((n, 64))
    try:
        for kernel in ("pipe", "tile", "rows"):
            ctx.set_tuning("spmm_kernel", kernel)
            out.fill_zero()
            D.spmm(ctx, csr, ctx.to_device(hb.x), ctx.to_device(bias), out, act="relu")
            assert rel_err(out.numpy(), ref) < TIGHT, kernel
            # deterministic: a second launch reproduces the first bit for bit
            out2 = ctx.zeros((n, 64))
            D.spmm(ctx, csr, ctx.to_device(hb.x), ctx.to_device(bias), out2, act="relu")
            assert np.array_equal(out.numpy(), out2.numpy()), kernel
        # the pipelined kernel at other widths (1, 2, 4, 8 slabs; slab groups of 1 and 4), without bias / activation
        ctx.set_tuning("spmm_kernel", "pipe")
        for f, sgk in ((32, 0), (128, 1), (256, 4), (256, 0)):
            ctx.set_tuning("spmm_sg", sgk)
            xf = rng.standard_normal((n, f), dtype=np.float32)
            o = ctx.zeros((n, f))
            D.spmm(ctx, csr, ctx.to_device(xf), None, o)
            assert rel_err(o.numpy(), _ref_spmm(hb, vals, xf, None, False)) < TIGHT, (f, sgk)
    finally:
        ctx.set_tuning("spmm_kernel", "auto")
        ctx.set_tuning("spmm_sg", 0)


def test_side_sections_order_and_capture(ctx):
    """gcnx_side_begin / _end / _join: a side section sees everything submitted before it, the main stream sees the
    side results after the join, d2h joins implicitly, and a captured sequence with a side section replays
    correctly (the graph gets two branches)."""
    from gcnx import device as D
    rng = np.random.default_rng(3)
    n, f = 20000, 64
    x = rng.standard_normal((n, f), dtype=np.float32)
    w = (rng.standard_normal((f, f)) / 8).astype(np.float32)
    dx, dw_, dh = ctx.to_device(x), ctx.to_device(w), ctx.empty((n, f))
    s1, s2, out = ctx.empty(f), ctx.empty(f), ctx.empty((n, f))

    def seq():
        D.gemm(ctx, dx, dw_, None, dh)                     # main: dh = x w
        with ctx.side():
            D.act_bias_grad(ctx, dh, None, dh, None, db=s1)  # side: column sums of dh (must see the finished GEMM)
        D.gemm(ctx, dh, dw_, None, out)                    # main, concurrently: out = dh w
        ctx.join()
        D.act_bias_grad(ctx, out, None, out, None, db=s2)  # main after the join

    ref_h = x.astype(np.float64) @ w.astype(np.float64)
    seq()
    assert rel_err(s1.numpy(), ref_h.sum(0)) < TIGHT          # numpy() = d2h, which joins the side stream
    assert rel_err(s2.numpy(), (ref_h @ w.astype(np.float64)).sum(0)) < TIGHT
    first = (s1.numpy().copy(), s2.numpy().copy())
    g = ctx.capture(seq)
    for _ in range(3):
        s1.fill_zero(); s2.fill_zero()
        g.launch()
        assert np.array_equal(s1.numpy(), first[0]) and np.array_equal(s2.numpy(), first[1])
    g.destroy()
    # misuse is reported, not silently accepted
    from gcnx import _lib
    assert ctx.lib.gcnx_side_end(ctx.h) == 1 and "not inside" in _lib.last_error(ctx.h)
    with ctx.side():
        assert ctx.lib.gcnx_side_begin(ctx.h) == 1


@pytest.mark.parametrize("n,f", [(20001, 200), (32, 256), (256, 130), (7, 2)])
def test_bn_moments_equals_stats_finalize_pairs(ctx, n, f):
    """gcnx_bn_moments (4 launches; ONE for a batch of at most 256 rows -- the post-MLP of GeneralGNN runs on one row per graph)
    is bit-identical to bn_stats + bn_finalize taken twice (6 launches), and matches numpy's two-pass moments; the moving
    statistics get the Keras momentum update."""
    from gcnx import device as D
    rng = np.random.default_rng(21)
    z = (rng.standard_normal((n, f)) * 3 + 50).astype(np.float32)     # large mean: the centred pass matters
    dz = ctx.to_device(z)
    mean, inv, sums = ctx.empty(f), ctx.empty(f), ctx.empty(2 * f)
    mm, mv = ctx.to_device(np.full(f, 0.5, np.float32)), ctx.to_device(np.full(f, 2.0, np.float32))
    D.bn_moments(ctx, dz, sums, mean, inv, mm, mv)
    m2, i2 = ctx.empty(f), ctx.empty(f)
    mm2, mv2 = ctx.to_device(np.full(f, 0.5, np.float32)), ctx.to_device(np.full(f, 2.0, np.float32))
    D.bn_stats(ctx, dz, sums)
    D.bn_finalize(ctx, sums, n, m2, i2)
    D.bn_stats(ctx, dz, sums, shift=m2)
    D.bn_finalize(ctx, sums, n, m2, i2, mm2, mv2, shift=m2)
    for a, b in ((mean, m2), (inv, i2), (mm, mm2), (mv, mv2)):
        assert np.array_equal(a.numpy(), b.numpy())
    z64 = z.astype(np.float64)
    assert rel_err(mean.numpy(), z64.mean(0)) < TIGHT
    assert rel_err(inv.numpy(), 1.0 / np.sqrt(z64.var(0) + 1e-3)) < TIGHT
    assert rel_err(mm.numpy(), 0.99 * 0.5 + 0.01 * z64.mean(0)) < TIGHT


@pytest.mark.parametrize("n,f,act", [(32, 256, "prelu"), (7, 2, None), (256, 130, "relu"), (3000, 96, "prelu")])
def test_bn_act_bwd_one_call_equals_stats_plus_apply_and_the_oracle(ctx, n, f, act):
    """gcnx_bn_act_bwd (ONE launch for a batch of at most 256 rows -- the post-MLP of GeneralGNN -- three otherwise) against its
    two halves gcnx_bn_act_bwd_stats + gcnx_bn_act_bwd_apply (the sync-BN form) bit for bit: dz, the column sums, dgamma, dbeta,
    dalpha; and against the oracle's batch-norm + activation backward in fp64.  In place (dz over dy) as the models call it."""
    from gcnx import device as D
    o = O()
    rng = np.random.default_rng(n + f)
    z = (rng.standard_normal((n, f)) * 2 + 1).astype(np.float32)
    dy = rng.standard_normal((n, f)).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(f)).astype(np.float32); beta = (0.3 * rng.standard_normal(f)).astype(np.float32)
    alpha = (0.25 * rng.random(f)).astype(np.float32) if act == "prelu" else None
    z64 = z.astype(np.float64)
    mean = z64.mean(0).astype(np.float32); inv = (1.0 / np.sqrt(z64.var(0) + 1e-3)).astype(np.float32)
    dev = lambda a: None if a is None else ctx.to_device(a)
    dm, di, dg, db_, da = dev(mean), dev(inv), dev(gamma), dev(beta), dev(alpha)
    dzv = dev(z)
    # one call, in place
    d1 = dev(dy); s1 = ctx.zeros(3 * f); gg1, gb1, ga1 = ctx.zeros(f), ctx.zeros(f), (ctx.zeros(f) if act == "prelu" else None)
    D.bn_act_bwd(ctx, d1, dzv, dm, di, dg, db_, d1, s1, act=act, alpha=da, training=True, dgamma=gg1, dbeta=gb1, dalpha=ga1)
    # the two halves
    d2 = dev(dy); s2 = ctx.zeros(3 * f); gg2, gb2, ga2 = ctx.zeros(f), ctx.zeros(f), (ctx.zeros(f) if act == "prelu" else None)
    D.bn_act_bwd_stats(ctx, d2, dzv, dm, di, dg, db_, s2, act=act, alpha=da, dgamma=gg2, dbeta=gb2, dalpha=ga2)
    D.bn_act_bwd_apply(ctx, d2, dzv, dm, di, dg, db_, s2, n, d2, act=act, alpha=da, training=True)
    assert np.array_equal(d1.numpy(), d2.numpy()) and np.array_equal(s1.numpy()[:2 * f], s2.numpy()[:2 * f])
    assert np.array_equal(gg1.numpy(), gg2.numpy()) and np.array_equal(gb1.numpy(), gb2.numpy())
    if act == "prelu":
        assert np.array_equal(ga1.numpy(), ga2.numpy())
    # oracle (fp64), with the same mean / inv
    xhat = (z64 - mean.astype(np.float64)) * inv.astype(np.float64)
    zb = gamma.astype(np.float64) * xhat + beta.astype(np.float64)
    dy64 = dy.astype(np.float64)
    dzb = o.act_bwd(dy64, zb, act, None if alpha is None else alpha.astype(np.float64))
    ref, rgam, rbet = o.bn_bwd(dzb, {"xhat": xhat, "inv": inv.astype(np.float64), "training": True}, gamma.astype(np.float64))
    assert rel_err(d1.numpy(), ref) < 2e-5 and rel_err(gg1.numpy(), rgam) < 2e-5 and rel_err(gb1.numpy(), rbet) < 2e-5
    if act == "prelu":
        assert rel_err(ga1.numpy(), (dy64 * np.minimum(zb, 0)).sum(0)) < 2e-5


def test_new_entry_points_zero_sizes_and_argument_errors(ctx):
    """Edge cases of the entry points added for the fused head, side sections, BN moments and the device collate:
    empty inputs are no-ops (or zero the outputs), bad arguments come back as GCNX_ERR_INVALID with a message."""
    from gcnx import _lib
    lib, h = ctx.lib, ctx.h
    buf = ctx.zeros((4, 4)); la = ctx.to_device(np.array([5.0, 5.0], np.float32)); dw = ctx.to_device(np.ones((4, 2), np.float32))
    # head: b == 0 -> loss_acc and dw zeroed, nothing else touched
    assert lib.gcnx_dense_softmax_cce(h, buf.ptr, 4, buf.ptr, None, buf.ptr, 0, 4, 2, 1.0, buf.ptr, la.ptr, dw.ptr, None, buf.ptr, 4, 1) == 0
    assert not la.numpy().any() and not dw.numpy().any()
    # head: too many classes / gradients without labels
    assert lib.gcnx_dense_softmax_cce(h, buf.ptr, 4, buf.ptr, None, buf.ptr, 4, 4, 33, 1.0, buf.ptr, la.ptr, None, None, None, 0, 1) == 1
    assert "at most 32 classes" in _lib.last_error(h)
    assert lib.gcnx_dense_softmax_cce(h, buf.ptr, 4, buf.ptr, None, None, 4, 4, 2, 1.0, buf.ptr, la.ptr, dw.ptr, None, buf.ptr, 4, 0) == 1
    assert "gradients need labels" in _lib.last_error(h)
    assert lib.gcnx_dense_softmax_cce(h, buf.ptr, 4, buf.ptr, None, buf.ptr, 4, 4, 2, 1.0, buf.ptr, la.ptr, None, None, None, 0, 7) == 1
    assert "unknown cce_mode" in _lib.last_error(h)
    # BN moments need rows
    assert lib.gcnx_bn_moments(h, buf.ptr, 4, 0, 4, 0.99, 1e-3, buf.ptr, buf.ptr, None, None) == 1
    # collate: b == 0 is a no-op; values in without values out is refused
    assert lib.gcnx_collate(h, None, 0, None, None, None, None, None, 0, 0, None, 0, None, None, None, None, 0, None, None, None) == 0
    ib = ctx.zeros(8, np.int32)
    assert lib.gcnx_collate(h, ib.ptr, 1, ib.ptr, ib.ptr, ib.ptr, buf.ptr, buf.ptr, 4, 4, None, 0, ib.ptr, ib.ptr, None, buf.ptr, 4,
                            None, ib.ptr, None) == 1
    assert "values in and out" in _lib.last_error(h)


def test_fused_step_entry_points_zero_sizes_and_argument_errors(ctx):
    """Edge cases of the single-stream step's entry points (pool+head, dense backward, deferred reductions, last
    gradient + SGD): empty inputs are no-ops that still leave defined outputs, bad arguments are GCNX_ERR_INVALID."""
    import ctypes as C
    from gcnx import _lib, device as D
    from gcnx.device import Segments
    lib, h = ctx.lib, ctx.h
    # pool + head with no graphs: loss and the head gradients are zeroed, db_relu too (through the fallback chain)
    buf = ctx.zeros((4, 4)); la = ctx.to_device(np.array([5.0, 5.0], np.float32)); dw = ctx.to_device(np.ones((4, 2), np.float32))
    dbr = ctx.to_device(np.ones(4, np.float32)); gp0 = ctx.zeros(1, np.int32)
    assert lib.gcnx_pool_dense_softmax_cce(h, gp0.ptr, buf.ptr, 4, 0, None, buf.ptr, 4, buf.ptr, None, buf.ptr, 0, 4, 2, 1.0, buf.ptr,
                                           la.ptr, dw.ptr, None, buf.ptr, 4, dbr.ptr, 1) == 0
    assert not la.numpy().any() and not dw.numpy().any() and not dbr.numpy().any()
    # db_relu needs the gradient outputs and SUM / AVG pooling
    assert lib.gcnx_pool_dense_softmax_cce(h, gp0.ptr, buf.ptr, 4, 2, None, buf.ptr, 4, buf.ptr, None, buf.ptr, 1, 4, 2, 1.0, buf.ptr,
                                           la.ptr, dw.ptr, None, buf.ptr, 4, dbr.ptr, 1) == 1
    assert "db_relu needs" in _lib.last_error(h)
    # dense backward with no rows: dW is zeroed, dX untouched; both outputs are required
    x0 = ctx.empty((0, 64)); dh0 = ctx.empty((0, 8)); w = ctx.zeros((64, 8)); dx0 = ctx.empty((0, 64))
    dwz = ctx.to_device(np.ones((64, 8), np.float32)); dbz = ctx.to_device(np.ones(64, np.float32))
    D.dense_bwd(ctx, x0, dh0, w, dx0, dwz, db_prev=dbz)
    assert not dwz.numpy().any() and not dbz.numpy().any()
    assert lib.gcnx_dense_bwd(h, None, 64, None, 8, w.ptr, 0, 64, 8, 0, None, 64, None, 0, None, dwz.ptr) == 1
    assert "dx and dw are both required" in _lib.last_error(h)
    # deferred form: scratch too small -> plain gcnx_dense_bwd, nothing pending; pending must not be NULL
    rng = np.random.default_rng(0)
    x = ctx.to_device(rng.standard_normal((300, 64), dtype=np.float32)); dh = ctx.to_device(rng.standard_normal((300, 8), dtype=np.float32))
    dx, dw1, dw2, db1, db2 = ctx.empty((300, 64)), ctx.empty((64, 8)), ctx.empty((64, 8)), ctx.empty(64), ctx.empty(64)
    pend = D.dense_bwd_deferred(ctx, x, dh, w, dx, dw1, ctx.empty(4), y_mask=x, db_prev=db1)
    assert pend.colpart is None and pend.slabs is None
    D.dense_bwd(ctx, x, dh, w, dx, dw2, y_mask=x, db_prev=db2)
    assert np.array_equal(dw1.numpy(), dw2.numpy()) and np.array_equal(db1.numpy(), db2.numpy())
    assert D.dense_bwd_scratch_floats(ctx, 300, 64, 8) > 0 and D.dense_bwd_scratch_floats(ctx, 300, 10, 8) == 0
    assert lib.gcnx_dense_bwd_deferred(h, x.ptr, 64, dh.ptr, 8, w.ptr, 300, 64, 8, 0, dx.ptr, 64, None, 0, None, dw1.ptr, None, 0, None) == 1
    assert "pending is NULL" in _lib.last_error(h)
    # last gradient + SGD: dW outside the flat gradient buffer, and a pending result outside it, are refused
    params, grads = ctx.zeros(64 * 8 + 10), ctx.zeros(64 * 8 + 10)
    with pytest.raises(_lib.GcnxError, match="inside the flat gradient buffer"):
        D.gemm_dw_sgd(ctx, x, dh, ctx.empty((64, 8)), params, grads, 0.1)
    bad = _lib.PendingReduce(colpart=x.ptr, crows=4, cf=64, cout=db1.ptr, slabs=None, total=0, nsplit=0, out=None)
    with pytest.raises(_lib.GcnxError, match="pending column sums must land"):
        D.gemm_dw_sgd(ctx, x, dh, grads.flat(0, 64 * 8, (64, 8)), params, grads, 0.1, pending=bad)
    # n == 0 with a valid layout: dW zeroed, every parameter still updated with its (other) gradients
    g0 = np.arange(64 * 8 + 10, dtype=np.float32); grads = ctx.to_device(g0); params = ctx.zeros(64 * 8 + 10)
    D.gemm_dw_sgd(ctx, x0, dh0, grads.flat(0, 64 * 8, (64, 8)), params, grads, 0.5)
    exp = g0.copy(); exp[:64 * 8] = 0
    assert np.array_equal(grads.numpy(), exp) and np.array_equal(params.numpy(), -0.5 * exp)


def test_device_collate_single_graph_and_repeated_selection(ctx):
    """gcnx_collate with a one-graph batch, and with the same graph selected twice (sampling with replacement):
    the copies are re-based independently."""
    from gcnx import Graph, ListDataset, synth, DeviceDataset
    from gcnx.device_loader import collate_on_device
    raw = synth.tiny_graphs(5, 8, seed=2)
    dds = DeviceDataset(ctx, ListDataset([Graph(x=x, a=a, y=y) for x, a, y in raw]), normalize="spektral")
    one = collate_on_device(dds, [3])
    n3 = raw[3][0].shape[0]
    assert one.n == n3 and one.n_graphs == 1 and np.array_equal(one.x.numpy(), raw[3][0].astype(np.float32))
    assert np.array_equal(one.seg.dev.numpy(), [0, n3])
    two = collate_on_device(dds, [3, 3])
    rp, ci = two.a.rowptr.numpy(), two.a.colidx.numpy()[:two.a.nnz]
    assert two.n == 2 * n3 and np.array_equal(two.x.numpy()[:n3], two.x.numpy()[n3:])
    e = rp[n3]
    assert np.array_equal(rp[n3:] - e, rp[:n3 + 1]) and np.array_equal(ci[e:] - n3, ci[:e])
    assert np.array_equal(two.a.vals.numpy()[:e], two.a.vals.numpy()[e:2 * e])


def test_spmm_empty_rows_single_nodes_and_strided_views(ctx):
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR
    # rows 0 and 3 have no entries; row 4 is a lone self-loop
    rowptr = np.array([0, 0, 2, 3, 3, 4], np.int32)
    colidx = np.array([0, 2, 1, 4], np.int32)
    hb = synth.HostBatch(np.arange(40, dtype=np.float32).reshape(5, 8), rowptr, colidx, None, np.array([0, 4, 5], np.int32),
                         np.eye(2, dtype=np.float32))
    a = DeviceCSR.from_host_csr(ctx, rowptr, colidx, None, hb.graph_ptr)
    out = ctx.zeros((5, 8))
    D.spmm(ctx, a, ctx.to_device(hb.x), None, out)
    assert np.array_equal(out.numpy(), _ref_spmm(hb, None, hb.x).astype(np.float32))
    # column-slice views of wider buffers (Spektral's concat-skip written in place)
    hb2 = synth.ecoli_batch(2, 32, seed=5)
    a2, vals = _csr(ctx, hb2, True)
    wide_in = ctx.to_device(np.concatenate([np.zeros_like(hb2.x), hb2.x], 1))       # h = cols 32..64
    wide_out = ctx.zeros((hb2.n, 96))
    D.spmm(ctx, a2, wide_in.cols(32, 64), None, wide_out.cols(64, 96))
    got = wide_out.numpy()
    assert rel_err(got[:, 64:], _ref_spmm(hb2, vals, hb2.x)) < TIGHT and not got[:, :64].any()


def test_spmm_zero_sizes_and_argument_errors(ctx):
    from gcnx import _lib
    lib = ctx.lib
    assert lib.gcnx_spmm_csr(ctx.h, None, None, None, None, 0, None, None, 0, 0, 16, 0, None) == 0
    buf = ctx.zeros((4, 4))
    rc = lib.gcnx_spmm_csr(ctx.h, buf.ptr, buf.ptr, None, buf.ptr, 2, None, buf.ptr, 4, 4, 4, 0, None)
    assert rc == 1 and "leading dimension" in _lib.last_error(ctx.h)          # ldh < f
    rc = lib.gcnx_spmm_csr(ctx.h, buf.ptr, buf.ptr, None, buf.ptr, 4, None, buf.ptr, 4, 4, 4, 0, None)
    assert rc == 1 and "in-place" in _lib.last_error(ctx.h)


def test_spmm_full_size_properties(ctx):
    """BASELINE config 3 size (1M nodes / 10M entries, F=256): size-independent properties.
    A*1 = row degree exactly (small integers are exact in fp32); linearity A(ah1+h2) = aAh1+Ah2."""
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR
    hb = synth.block_diag_batch(with_x=False)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    n, f = hb.n, 256
    ones = ctx.to_device(np.ones((n, f), np.float32))
    out = ctx.empty((n, f))
    D.spmm(ctx, a, ones, None, out)
    got = out.numpy()
    deg = np.diff(hb.rowptr).astype(np.float32)
    assert np.array_equal(got[:, 0], deg) and np.array_equal(got[:, 255], deg) and np.array_equal(got[::997].T[7], deg[::997])
    rng = np.random.default_rng(0)
    h1 = rng.standard_normal((n, f), dtype=np.float32); h2 = rng.standard_normal((n, f), dtype=np.float32)
    d1, d2 = ctx.to_device(h1), ctx.to_device(h2)
    o1, o2, o3 = ctx.empty((n, f)), ctx.empty((n, f)), ctx.empty((n, f))
    D.spmm(ctx, a, d1, None, o1); D.spmm(ctx, a, d2, None, o2)
    d3 = ctx.to_device(np.float32(0.5) * h1 + h2)
    D.spmm(ctx, a, d3, None, o3)
    lin = np.float32(0.5) * o1.numpy() + o2.numpy()
    assert rel_err(o3.numpy(), lin) < 1e-5
    # spot-check 2000 random rows against the oracle definition
    rows = rng.integers(0, n, 2000)
    ref = np.stack([h1[hb.colidx[hb.rowptr[r]:hb.rowptr[r + 1]]].astype(np.float64).sum(0) for r in rows])
    assert rel_err(o1.numpy()[rows], ref) < TIGHT


@pytest.mark.parametrize("mode", ["max", "min"])
def test_spmm_minmax_aggregation_and_its_gradient_with_ties(ctx, mode):
    """gcnx_spmm_csr_minmax / _bwd (GeneralConv(aggregate="max" | "min"), tf.math.unsorted_segment_max / _min): values, tie counts
    and the gradient -- every message equal to the extremum gets dy / count -- on integer-valued messages (exact ties), a ragged
    width, rows without entries (identity value, no gradient), against the oracle's restatement (which torch.amax / amin pin,
    tests/test_oracle.py); the gradient runs over the transposed operator of a DIRECTED pattern."""
    import scipy.sparse as sp
    from gcnx import device as D
    from gcnx.device import DeviceCSR
    o = O()
    rng = np.random.default_rng(17)
    n, f = 301, 37
    m = sp.random(n, n, density=0.03, random_state=3, format="csr"); m.data[:] = 1.0
    m = m.tolil(); m[7, :] = 0; m[200, :] = 0; m = m.tocsr(); m.eliminate_zeros(); m.sort_indices()
    rp, ci = m.indptr.astype(np.int32), m.indices.astype(np.int32)
    a = DeviceCSR.from_host_csr(ctx, rp, ci, None, None)
    h = rng.integers(-3, 4, (n, f)).astype(np.float32)
    dy = rng.standard_normal((n, f), dtype=np.float32)
    out, cnt, dh = ctx.empty((n, f)), ctx.empty((n, f)), ctx.empty((n, f))
    dh_ = ctx.to_device(h)
    D.spmm_minmax(ctx, a, dh_, out, cnt, mode)
    rout, rcnt = o.aggregate_minmax(rp.astype(np.int64), ci.astype(np.int64), h.astype(np.float64), mode)
    assert np.array_equal(out.numpy(), rout.astype(np.float32)) and np.array_equal(cnt.numpy(), rcnt.astype(np.float32))
    assert rcnt.max() >= 2 and out.numpy()[7, 0] == (np.finfo(np.float32).max if mode == "min" else -np.finfo(np.float32).max)
    D.spmm_minmax_bwd(ctx, a.transpose(), dh_, out, cnt, ctx.to_device(dy), dh)
    rdh = o.aggregate_minmax_bwd(rp.astype(np.int64), ci.astype(np.int64), h.astype(np.float64), rout, rcnt, dy.astype(np.float64))
    assert rel_err(dh.numpy(), rdh) < TIGHT
    dh2 = ctx.empty((n, f)); D.spmm_minmax_bwd(ctx, a.transpose(), dh_, out, cnt, ctx.to_device(dy), dh2)
    assert np.array_equal(dh.numpy(), dh2.numpy())                                    # deterministic


def test_spmm_prod_aggregation_and_its_zero_aware_gradient(ctx):
    """gcnx_spmm_csr_prod / _bwd (GeneralConv(aggregate="prod"), tf.math.unsorted_segment_prod; r4): products, the auxiliary
    array and the gradient -- prod / message, the product of the others for the only zero of a row, nothing with two or more zeros --
    on integer-valued messages (exact zeros, exact products), a ragged width, rows without entries (1, no gradient), against the
    oracle's restatement (which torch.prod pins, tests/test_oracle.py); the gradient runs over the transposed operator of a
    DIRECTED pattern; bit-reproducible."""
    import scipy.sparse as sp
    from gcnx import device as D
    from gcnx.device import DeviceCSR
    o = O()
    rng = np.random.default_rng(19)
    n, f = 301, 37
    m = sp.random(n, n, density=0.02, random_state=5, format="csr"); m.data[:] = 1.0
    m = m.tolil(); m[7, :] = 0; m[200, :] = 0; m = m.tocsr(); m.eliminate_zeros(); m.sort_indices()
    rp, ci = m.indptr.astype(np.int32), m.indices.astype(np.int32)
    a = DeviceCSR.from_host_csr(ctx, rp, ci, None, None)
    h = rng.integers(-2, 3, (n, f)).astype(np.float32)
    dy = rng.standard_normal((n, f), dtype=np.float32)
    out, aux, dh = ctx.empty((n, f)), ctx.empty((n, f)), ctx.empty((n, f))
    dh_ = ctx.to_device(h)
    D.spmm_minmax(ctx, a, dh_, out, aux, "prod")
    rout, raux = o.aggregate_prod(rp.astype(np.int64), ci.astype(np.int64), h.astype(np.float64))
    assert np.array_equal(out.numpy(), rout.astype(np.float32)) and np.array_equal(aux.numpy(), raux.astype(np.float32))
    assert np.all(out.numpy()[7] == 1.0) and (rout == 0).any() and (raux != rout).any()
    D.spmm_minmax_bwd(ctx, a.transpose(), dh_, out, aux, ctx.to_device(dy), dh, "prod")
    rdh = o.aggregate_prod_bwd(rp.astype(np.int64), ci.astype(np.int64), h.astype(np.float64), rout, raux, dy.astype(np.float64))
    assert rel_err(dh.numpy(), rdh) < TIGHT
    dh2 = ctx.empty((n, f)); D.spmm_minmax_bwd(ctx, a.transpose(), dh_, out, aux, ctx.to_device(dy), dh2, "prod")
    assert np.array_equal(dh.numpy(), dh2.numpy())                                    # deterministic


def test_spmm_balanced_deal_of_the_tile_graphs_changes_no_bit(ctx):
    """The plan's balanced deal (r3, csrc/spmm.hip balance_tile_list; knob spmm_bal): with >= 1.5 tile graphs per CU the
    graph list is laid out so that the kernel's static snake deal gives every workgroup about the same cost instead of the
    plain size order.  Only WHERE a unit runs changes: forward aggregation (bias + ReLU), the unweighted operator and the
    folded pool backward are bit-identical with the deal on and off, and equal the oracle definition."""
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR, Segments
    hb = synth.block_diag_batch(130_000, 1_000_000, 64, seed=11, mean_size=200, with_x=False)
    assert hb.n_graphs >= 2 * 256
    rng = np.random.default_rng(12)
    f = 64
    h = rng.standard_normal((hb.n, f), dtype=np.float32); bias = rng.standard_normal(f).astype(np.float32)
    dp = rng.standard_normal((hb.n_graphs, f), dtype=np.float32)
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    res = {}
    try:
        for bal in (0, 1, 30045):
            ctx.set_tuning("spmm_bal", bal)
            a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)      # a new plan under this setting
            assert a.plan is not None
            o1, o2, o3 = ctx.zeros((hb.n, f)), ctx.zeros((hb.n, f)), ctx.zeros((hb.n, f))
            D.spmm(ctx, a, ctx.to_device(h), ctx.to_device(bias), o1, act="relu")
            D.spmm(ctx, a.unweighted(), ctx.to_device(h), None, o2)
            D.spmm_pool_bwd(ctx, a, ctx.to_device(np.maximum(h, 0)), Segments(ctx, hb.graph_ptr), ctx.to_device(dp), o3, "sum")
            res[bal] = (o1.numpy(), o2.numpy(), o3.numpy())
    finally:
        ctx.set_tuning("spmm_bal", 1)
    for bal in (1, 30045):
        for x, y in zip(res[0], res[bal]):
            assert np.array_equal(x, y)
    assert rel_err(res[1][0], _ref_spmm(hb, vals, h, bias, relu=True)) < TIGHT
    assert rel_err(res[1][1], _ref_spmm(hb, None, h)) < TIGHT


@pytest.mark.parametrize("n,fi,fo", [(1000, 128, 128), (77, 10, 6), (4096, 256, 256), (333, 32, 2), (65, 130, 70),
                                     (32, 1280, 256), (5, 256, 2), (64, 67, 130), (1, 64, 1)])     # (few rows: the thin kernel)
def test_gemm_forward_parity(ctx, n, fi, fo):
    from gcnx import device as D
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, fi), dtype=np.float32); w = (rng.standard_normal((fi, fo)) / np.sqrt(fi)).astype(np.float32)
    b = rng.standard_normal(fo).astype(np.float32); al = rng.random(fo).astype(np.float32)
    dx, dw, db, dal = (ctx.to_device(v) for v in (x, w, b, al))
    out = ctx.empty((n, fo))
    z = x.astype(np.float64) @ w.astype(np.float64)
    D.gemm(ctx, dx, dw, None, out)
    assert rel_err(out.numpy(), z) < TIGHT
    D.gemm(ctx, dx, dw, db, out, act="relu")
    assert rel_err(out.numpy(), np.maximum(z + b, 0)) < TIGHT
    D.gemm(ctx, dx, dw, db, out, act="prelu", alpha=dal)
    zb = z + b
    assert rel_err(out.numpy(), np.maximum(zb, 0) + al * np.minimum(zb, 0)) < TIGHT


def test_gemm_backward_parity_and_fusions(ctx):
    from gcnx import device as D
    rng = np.random.default_rng(1)
    # db: (3000, 128) / (50000, 64) one-launch reduce of the epilogue's partial rows, (140001, 64) its two-stage form,
    # (129, 10) the column-sum pass over dX (ragged width)
    for n, fi, fo in ((3000, 128, 128), (50000, 64, 96), (129, 10, 6), (140001, 64, 32)):
        x = rng.standard_normal((n, fi), dtype=np.float32); dh = rng.standard_normal((n, fo), dtype=np.float32)
        w = rng.standard_normal((fi, fo), dtype=np.float32); ymask = rng.standard_normal((n, fi), dtype=np.float32)
        d_x, d_dh, d_w, d_m = (ctx.to_device(v) for v in (x, dh, w, ymask))
        dw = ctx.empty((fi, fo))
        D.gemm_dw(ctx, d_x, d_dh, dw)                                     # split-K over the N rows
        assert rel_err(dw.numpy(), x.astype(np.float64).T @ dh.astype(np.float64)) < TIGHT
        dx = ctx.empty((n, fi)); db = ctx.empty(fi)
        ref = dh.astype(np.float64) @ w.astype(np.float64).T
        D.gemm_dx(ctx, d_dh, d_w, dx)
        assert rel_err(dx.numpy(), ref) < TIGHT
        D.gemm_dx(ctx, d_dh, d_w, dx, y_mask=d_m, db=db)                  # fused ReLU mask + BiasAddGrad
        refm = ref * (ymask > 0)
        assert rel_err(dx.numpy(), refm) < TIGHT and rel_err(db.numpy(), refm.sum(0)) < TIGHT
        db2 = ctx.empty(fi)
        D.gemm_dx(ctx, d_dh, d_w, dx, y_mask=d_m, db=db2)
        assert np.array_equal(db.numpy(), db2.numpy())                     # fixed summation order
        base = rng.standard_normal((n, fi), dtype=np.float32)
        acc = ctx.to_device(base)
        D.gemm_dx(ctx, d_dh, d_w, acc, accumulate=True)                   # skip-connection gradient add
        assert rel_err(acc.numpy(), ref + base) < TIGHT
    # split-K is deterministic: two runs are bitwise identical
    a1 = ctx.empty((64, 96)); a2 = ctx.empty((64, 96))
    D.gemm_dw(ctx, d_x, d_dh, dw)
    x = rng.standard_normal((50000, 64), dtype=np.float32); dh = rng.standard_normal((50000, 96), dtype=np.float32)
    d_x, d_dh = ctx.to_device(x), ctx.to_device(dh)
    D.gemm_dw(ctx, d_x, d_dh, a1); D.gemm_dw(ctx, d_x, d_dh, a2)
    assert np.array_equal(a1.numpy(), a2.numpy())


@pytest.mark.parametrize("n,fi,fo,prec", [(20498, 128, 128, "f32"), (3000, 64, 96, "f32"), (70, 128, 4, "f32"),
                                          (129, 10, 6, "f32"), (1000, 128, 128, "bf16x3")])
def test_dense_bwd_equals_gemm_dx_plus_gemm_dw(ctx, n, fi, fo, prec):
    """gcnx_dense_bwd (dX tiles + dW split-K tiles in one launch, both reductions in a second) against the oracle
    products and against the separate calls; ragged / bf16 shapes take the two calls inside."""
    from gcnx import device as D
    rng = np.random.default_rng(n + fo)
    x = rng.standard_normal((n, fi), dtype=np.float32); dh = rng.standard_normal((n, fo), dtype=np.float32)
    w = rng.standard_normal((fi, fo), dtype=np.float32)
    d_x, d_dh, d_w = ctx.to_device(x), ctx.to_device(dh), ctx.to_device(w)
    tol = TIGHT if prec == "f32" else TOL
    for masked in (True, False):
        dx, dw, db = ctx.empty((n, fi)), ctx.empty((fi, fo)), ctx.empty(fi)
        D.dense_bwd(ctx, d_x, d_dh, d_w, dx, dw, prec=prec, y_mask=d_x if masked else None, db_prev=db if masked else None)
        rdx = dh.astype(np.float64) @ w.astype(np.float64).T * ((x > 0) if masked else 1.0)
        assert rel_err(dx.numpy(), rdx) < tol
        assert rel_err(dw.numpy(), x.astype(np.float64).T @ dh.astype(np.float64)) < tol
        if masked:
            assert rel_err(db.numpy(), rdx.sum(0)) < tol
        dx2, dw2, db2 = ctx.empty((n, fi)), ctx.empty((fi, fo)), ctx.empty(fi)
        D.gemm_dx(ctx, d_dh, d_w, dx2, prec=prec, y_mask=d_x if masked else None, db=db2 if masked else None)
        D.gemm_dw(ctx, d_x, d_dh, dw2, prec=prec)
        assert np.array_equal(dx.numpy(), dx2.numpy())                 # same tile arithmetic
        assert rel_err(dw.numpy(), dw2.numpy()) < TIGHT               # (split-K slices differ: summation order)
        dx3, dw3, db3 = ctx.empty((n, fi)), ctx.empty((fi, fo)), ctx.empty(fi)
        D.dense_bwd(ctx, d_x, d_dh, d_w, dx3, dw3, prec=prec, y_mask=d_x if masked else None, db_prev=db3 if masked else None)
        assert np.array_equal(dw.numpy(), dw3.numpy()) and (not masked or np.array_equal(db.numpy(), db3.numpy()))


@pytest.mark.parametrize("n,fi,fo,prec", [(20498, 128, 128, "f32"), (3000, 64, 96, "f32"), (10, 6, 4, "f32"),
                                          (129, 10, 6, "f32"), (1000, 128, 128, "bf16x3")])
def test_gemm_dw_sgd_equals_gemm_dw_then_sgd_bitwise(ctx, n, fi, fo, prec):
    """gcnx_gemm_dw_sgd: dW lands in the middle of a flat gradient buffer and every parameter is updated; the same
    bits as gcnx_gemm_dw + gcnx_sgd (same split-K slices, same update arithmetic)."""
    from gcnx import device as D
    rng = np.random.default_rng(fi * fo)
    x = ctx.to_device(rng.standard_normal((n, fi), dtype=np.float32))
    dh = ctx.to_device(rng.standard_normal((n, fo), dtype=np.float32))
    before, off, tail = 300, 300, 517                        # parameters in front of and behind dW
    n_params = before + fi * fo + tail
    p0 = rng.standard_normal(n_params).astype(np.float32); g0 = rng.standard_normal(n_params).astype(np.float32)
    out = []
    for fused in (False, True):
        params, grads = ctx.to_device(p0), ctx.to_device(g0)
        dw = grads.flat(off, fi * fo, (fi, fo))
        if fused:
            D.gemm_dw_sgd(ctx, x, dh, dw, params, grads, 0.05, prec=prec)
        else:
            D.gemm_dw(ctx, x, dh, dw, prec=prec)
            D.sgd(ctx, params, grads, 0.05)
        out.append((params.numpy(), grads.numpy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert not np.array_equal(out[1][0], p0)
    from gcnx._lib import GcnxError
    with pytest.raises(GcnxError, match="inside the flat gradient buffer"):
        D.gemm_dw_sgd(ctx, x, dh, ctx.empty((fi, fo)), ctx.to_device(p0), ctx.to_device(g0), 0.05, prec=prec)


@pytest.mark.parametrize("n,fi,fo,prec", [(20498, 128, 128, "f32"), (3000, 64, 96, "f32"), (129, 10, 6, "f32"),
                                          (1000, 128, 128, "bf16x3")])
def test_dense_bwd_deferred_then_gemm_dw_sgd_equals_the_separate_calls(ctx, n, fi, fo, prec):
    """gcnx_dense_bwd_deferred leaves db_prev / dW of one layer unreduced in the caller's scratch; gcnx_gemm_dw_sgd
    (pending=...) finishes them, reduces its own dW and updates every parameter -- the same bits as gcnx_dense_bwd +
    gcnx_gemm_dw + gcnx_sgd.  Ragged / bf16 shapes: the deferred call degenerates to gcnx_dense_bwd (pending empty)."""
    from gcnx import device as D
    rng = np.random.default_rng(n + fi)
    x1 = ctx.to_device(rng.standard_normal((n, fi), dtype=np.float32))       # layer input (also the ReLU mask source)
    dh = ctx.to_device(rng.standard_normal((n, fo), dtype=np.float32))
    w = ctx.to_device(rng.standard_normal((fi, fo), dtype=np.float32))
    x0 = ctx.to_device(rng.standard_normal((n, 64), dtype=np.float32))       # the previous layer's input
    # flat buffers: [w0 (64 x fi) | b_prev (fi) | w (fi x fo) | tail]
    o_w0, o_b, o_w = 0, 64 * fi, 64 * fi + ((fi + 3) // 4) * 4
    n_params = o_w + fi * fo + 37
    p0 = rng.standard_normal(n_params).astype(np.float32); g0 = rng.standard_normal(n_params).astype(np.float32)
    out = []
    for deferred in (False, True):
        params, grads = ctx.to_device(p0), ctx.to_device(g0)
        gw0, gb, gw = grads.flat(o_w0, 64 * fi, (64, fi)), grads.flat(o_b, fi), grads.flat(o_w, fi * fo, (fi, fo))
        dx = ctx.empty((n, fi))
        if deferred:
            scratch = ctx.empty(max(D.dense_bwd_scratch_floats(ctx, n, fi, fo), 4))
            pend = D.dense_bwd_deferred(ctx, x1, dh, w, dx, gw, scratch, prec=prec, y_mask=x1, db_prev=gb)
            assert (pend.colpart is not None) == (prec == "f32" and fi % 64 == 0)
            D.gemm_dw_sgd(ctx, x0, dx, gw0, params, grads, 0.05, prec=prec, pending=pend)
        else:
            D.dense_bwd(ctx, x1, dh, w, dx, gw, prec=prec, y_mask=x1, db_prev=gb)
            D.gemm_dw(ctx, x0, dx, gw0, prec=prec)
            D.sgd(ctx, params, grads, 0.05)
        out.append((params.numpy(), grads.numpy(), dx.numpy()))
    for a0, a1 in zip(out[0], out[1]):
        assert np.array_equal(a0, a1)
    assert not np.array_equal(out[1][0], p0)


def _bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 on the host (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


@pytest.mark.parametrize("n,fi,fo", [(1000, 128, 128), (77, 10, 6), (4096, 256, 256), (333, 32, 2), (300, 130, 70)])
def test_gemm_bf16_mfma_paths(ctx, n, fi, fo):
    """The bf16 MFMA weight GEMM of north_star.  bf16x3 (hi/lo split, 3 passes) is held to the
    1e-4 fp32 bar; plain bf16 is exact against an oracle fed bf16-rounded operands (the products
    are exact in fp32, only the accumulation order differs) and within 2e-2 of the fp32 result."""
    from gcnx import device as D
    rng = np.random.default_rng(n + fi)
    x = rng.standard_normal((n, fi), dtype=np.float32); w = (rng.standard_normal((fi, fo)) / np.sqrt(fi)).astype(np.float32)
    dh = rng.standard_normal((n, fo), dtype=np.float32); b = rng.standard_normal(fo).astype(np.float32)
    ym = rng.standard_normal((n, fi), dtype=np.float32)
    d_x, d_w, d_dh, d_b, d_ym = (ctx.to_device(v) for v in (x, w, dh, b, ym))
    x64, w64, dh64 = x.astype(np.float64), w.astype(np.float64), dh.astype(np.float64)
    xb, wb, dhb = (_bf16_round(v).astype(np.float64) for v in (x, w, dh))
    out = ctx.empty((n, fo)); dx = ctx.empty((n, fi)); dw = ctx.empty((fi, fo)); db = ctx.empty(fi)
    for prec, tol, (ax, aw, adh) in (("bf16x3", TOL, (x64, w64, dh64)), ("bf16", 2e-5, (xb, wb, dhb))):
        D.gemm(ctx, d_x, d_w, d_b, out, act="relu", prec=prec)
        assert rel_err(out.numpy(), np.maximum(ax @ aw + b, 0)) < tol, prec
        D.gemm_dx(ctx, d_dh, d_w, dx, prec=prec, y_mask=d_ym, db=db)
        ref = (adh @ aw.T) * (ym > 0)
        assert rel_err(dx.numpy(), ref) < tol and rel_err(db.numpy(), ref.sum(0)) < max(tol, 1e-5), prec
        D.gemm_dw(ctx, d_x, d_dh, dw, prec=prec)
        assert rel_err(dw.numpy(), ax.T @ adh) < tol, prec
    D.gemm(ctx, d_x, d_w, None, out, prec="bf16")
    assert rel_err(out.numpy(), x64 @ w64) < 2e-2


@pytest.mark.parametrize("n,fi,fo", [(5003, 256, 256), (2100, 16, 256), (9000, 128, 192), (40000, 64, 256)])
def test_gemm_f32_rowtile_kernel(ctx, n, fi, fo):
    """The fp32 row-tile kernel (weight slice in registers, 16x16x4 MFMA; >= 2048 rows, K in {16 .. 256}, 192 - 256
    output columns): X W with bias / ReLU / PReLU, dH W^T (K = fo, columns = fi: served when fi >= 192) with the ReLU
    mask, accumulate and the fused column sums; ragged last tile; column-slice views; against float64, and equal to
    the tiled kernel to rounding."""
    from gcnx import device as D
    rng = np.random.default_rng(n + fi + fo)
    x = rng.standard_normal((n, fi), dtype=np.float32); w = (rng.standard_normal((fi, fo)) / np.sqrt(fi)).astype(np.float32)
    b = rng.standard_normal(fo).astype(np.float32); al = rng.random(fo).astype(np.float32)
    d_x, d_w, d_b, d_al = (ctx.to_device(v) for v in (x, w, b, al))
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    out = ctx.empty((n, fo))
    z = x64 @ w64 + b
    D.gemm(ctx, d_x, d_w, d_b, out, act="relu")
    got = out.numpy()
    assert rel_err(got, np.maximum(z, 0)) < TIGHT
    D.gemm(ctx, d_x, d_w, d_b, out, act="prelu", alpha=d_al)
    assert rel_err(out.numpy(), np.where(z > 0, z, al * z)) < TIGHT
    D.gemm(ctx, d_x, d_w, None, out)
    assert rel_err(out.numpy(), x64 @ w64) < TIGHT
    ctx.set_tuning("gemm_stream", 0)                                   # the 64 x 64-tile kernel
    try:
        D.gemm(ctx, d_x, d_w, d_b, out, act="relu")
    finally:
        ctx.set_tuning("gemm_stream", 1)
    assert rel_err(out.numpy(), got) < TIGHT
    # dH W^T: here dH is [n, fo'] with fo' = fi of this case and W [fo, fi] -> dX [n, fo]
    w2 = (rng.standard_normal((fo, fi)) / np.sqrt(fi)).astype(np.float32)
    ym = rng.standard_normal((n, fo), dtype=np.float32)
    d_w2, d_ym = ctx.to_device(w2), ctx.to_device(ym)
    dx = ctx.zeros((n, fo)); db = ctx.empty(fo)
    ref = (x64 @ w2.astype(np.float64).T) * (ym > 0)
    D.gemm_dx(ctx, d_x, d_w2, dx, y_mask=d_ym, db=db)
    assert rel_err(dx.numpy(), ref) < TIGHT and rel_err(db.numpy(), ref.sum(0)) < TIGHT
    first = (dx.numpy().copy(), db.numpy().copy())
    D.gemm_dx(ctx, d_x, d_w2, dx, y_mask=d_ym, db=db)
    assert np.array_equal(dx.numpy(), first[0]) and np.array_equal(db.numpy(), first[1])       # fixed-order sums
    base = rng.standard_normal((n, fo), dtype=np.float32)
    dx.copy_from_host(base)
    D.gemm_dx(ctx, d_x, d_w2, dx, accumulate=True)                     # skip-connection gradients add up in place
    assert rel_err(dx.numpy(), base + x64 @ w2.astype(np.float64).T) < TIGHT
    wide_in, wide_out = ctx.to_device(np.concatenate([x, x], 1)), ctx.zeros((n, fo + 64))
    D.gemm(ctx, wide_in.cols(fi, 2 * fi), d_w, d_b, wide_out.cols(64, 64 + fo))
    wo = wide_out.numpy()
    assert rel_err(wo[:, 64:], z) < TIGHT and not wo[:, :64].any()


@pytest.mark.parametrize("n,fi,fo", [(40000, 256, 256), (33001, 256, 192), (70000, 128, 128), (36000, 128, 256), (50000, 256, 64)])
def test_gemm_bf16_streaming_kernel(ctx, n, fi, fo, monkeypatch):
    """The streaming form of the bf16 weight GEMM (csrc/gemm_stream.hip: weight planes resident in LDS, activation
    rows loaded straight into MFMA operand layout four K steps ahead, float4 epilogue), taken for tall activations
    (>= 32 768 rows, K = 128 / 256): X W + b with ReLU / PReLU and dH W^T with the fused ReLU mask, ragged row counts,
    one and two column halves, a partial second half -- bf16x3 at the fp32 bar, plain bf16 exact against
    bf16-rounded operands; and bit-identical to nothing less than itself: two runs agree."""
    import gcnx
    from gcnx import device as D
    rng = np.random.default_rng(n + fi + fo)
    x = rng.standard_normal((n, fi), dtype=np.float32); w = (rng.standard_normal((fi, fo)) / np.sqrt(fi)).astype(np.float32)
    dh = rng.standard_normal((n, fo), dtype=np.float32); b = rng.standard_normal(fo).astype(np.float32)
    al = rng.random(fo).astype(np.float32)
    ym = rng.standard_normal((n, fi), dtype=np.float32)
    d_x, d_w, d_dh, d_b, d_ym, d_al = (ctx.to_device(v) for v in (x, w, dh, b, ym, al))
    x64, w64, dh64 = x.astype(np.float64), w.astype(np.float64), dh.astype(np.float64)
    xb, wb, dhb = (_bf16_round(v).astype(np.float64) for v in (x, w, dh))
    out = ctx.empty((n, fo)); dx = ctx.empty((n, fi)); db = ctx.empty(fi)
    for prec, tol, (ax, aw, adh) in (("bf16x3", TOL, (x64, w64, dh64)), ("bf16", 2e-5, (xb, wb, dhb))):
        out.fill_zero()
        D.gemm(ctx, d_x, d_w, d_b, out, act="relu", prec=prec)
        got = out.numpy()
        assert rel_err(got, np.maximum(ax @ aw + b, 0)) < tol, prec
        D.gemm(ctx, d_x, d_w, d_b, out, act="relu", prec=prec)
        assert np.array_equal(out.numpy(), got)
        D.gemm(ctx, d_x, d_w, d_b, out, act="prelu", alpha=d_al, prec=prec)
        z = ax @ aw + b
        assert rel_err(out.numpy(), np.where(z > 0, z, al * z)) < tol, prec
        D.gemm(ctx, d_x, d_w, None, out, prec=prec)
        assert rel_err(out.numpy(), ax @ aw) < tol, prec
        dx.fill_zero()
        D.gemm_dx(ctx, d_dh, d_w, dx, prec=prec, y_mask=d_ym, db=db)
        ref = (adh @ aw.T) * (ym > 0)
        assert rel_err(dx.numpy(), ref) < tol and rel_err(db.numpy(), ref.sum(0)) < max(tol, 1e-5), prec
        D.gemm_dx(ctx, d_dh, d_w, dx, prec=prec)
        assert rel_err(dx.numpy(), adh @ aw.T) < tol, prec
        # X^T dH (fi = fo = 256: one split-K slice per CU with the whole product in registers, transposing LDS reads)
        dwv = ctx.empty((fi, fo))
        D.gemm_dw(ctx, d_x, d_dh, dwv, prec=prec)
        g1 = dwv.numpy()
        assert rel_err(g1, ax.T @ adh) < tol, prec
        D.gemm_dw(ctx, d_x, d_dh, dwv, prec=prec)
        assert np.array_equal(dwv.numpy(), g1)
    # column-slice views (leading dimension larger than the width) on both sides
    wide_in, wide_out = ctx.to_device(np.concatenate([x, x], 1)), ctx.zeros((n, fo + 64))
    D.gemm(ctx, wide_in.cols(fi, 2 * fi), d_w, d_b, wide_out.cols(64, 64 + fo), prec="bf16x3")
    wo = wide_out.numpy()
    assert rel_err(wo[:, 64:], x64 @ w64 + b) < TOL and not wo[:, :64].any()
    # the knob that keeps the tiled kernel gives the same result to rounding
    monkeypatch.setenv("GCNX_GEMM_STREAM", "0")
    c2 = gcnx.Context(0)
    try:
        o2 = c2.empty((n, fo))
        D.gemm(c2, c2.to_device(x), c2.to_device(w), c2.to_device(b), o2, act="relu", prec="bf16x3")
        assert rel_err(o2.numpy(), np.maximum(x64 @ w64 + b, 0)) < TOL
    finally:
        c2.close()


@pytest.mark.parametrize("f", [64, 128, 256])
@pytest.mark.parametrize("weighted", [False, True])
def test_spmm_bf16_features(ctx, f, weighted):
    """gcnx_spmm_csr_bf16 (bf16 in / bf16 out, fp32 accumulation) against the oracle's exact aggregation of the same bf16
    values: every output within one bf16 unit in the last place of the correctly rounded result (the device sums in fp32,
    in another order), the large majority identical; ragged last tile; the converters bit-exact (round to nearest even)."""
    from gcnx import device as D, synth
    o = O()
    hb = synth.ecoli_batch(3, f, seed=f)
    a, vals = _csr(ctx, hb, weighted)
    rng = np.random.default_rng(f)
    hf = rng.standard_normal((hb.n, f), dtype=np.float32)
    hf[0, :4] = [1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -1.0 - 2.0 ** -8, 0.0]          # ties
    bias = rng.standard_normal(f).astype(np.float32)
    h16 = D.to_bf16(ctx, ctx.to_device(hf))
    assert np.array_equal(h16.numpy(), o.bf16_bits(hf))
    assert np.array_equal(D.from_bf16(ctx, h16).numpy(), o.bf16_from_bits(o.bf16_bits(hf)))
    out = ctx.empty((hb.n, f), np.uint16)
    for relu in (True, False):
        D.spmm_bf16(ctx, a, h16, ctx.to_device(bias), out, act="relu" if relu else None)
        ref = o.spmm_csr_bf16(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals, h16.numpy(), bias, relu)
        want = o.bf16_bits(ref.astype(np.float32))
        got = out.numpy()
        # the final rounding is worth half a unit in the last place of the result (2^-9 relative), the fp32 accumulation
        # 2^-24 per operation relative to the sum of the terms' MAGNITUDES (a sum that cancels keeps that error)
        mag = o.spmm_csr_bf16(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None if vals is None else np.abs(vals),
                              o.bf16_bits(np.abs(o.bf16_from_bits(h16.numpy()))), np.abs(bias))
        vg = o.bf16_from_bits(got).astype(np.float64)
        assert np.all(np.abs(vg - ref) <= 2.0 ** -8 * np.abs(ref) + 2.0 ** -19 * mag)
        assert (got == want).mean() > 0.97                                        # and the large majority bit-identical
        if relu:
            assert not (got & 0x8000).any()
    D.spmm_bf16(ctx, a, h16, None, out)                                           # no bias
    ref = o.spmm_csr_bf16(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals, h16.numpy())
    assert (np.abs(out.numpy().astype(np.int32) - o.bf16_bits(ref.astype(np.float32)).astype(np.int32)) <= 1).mean() > 0.999


def test_spmm_bf16_long_rows_and_refusals(ctx):
    """Rows longer than the tile's LDS staging (the overflow path), and widths the kernel does not serve."""
    from gcnx import device as D, synth
    from gcnx._lib import GcnxError
    o = O()
    hb = synth.power_law_batch(n_graphs=1, graph_size=8192, f=64, seed=3)
    a, vals = _csr(ctx, hb, True)
    assert np.diff(hb.rowptr).max() > 1024
    h16 = D.to_bf16(ctx, ctx.to_device(hb.x))
    out = ctx.empty((hb.n, 64), np.uint16)
    D.spmm_bf16(ctx, a, h16, None, out)
    ref = o.spmm_csr_bf16(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals, h16.numpy())
    got = o.bf16_from_bits(out.numpy()).astype(np.float64)
    mag = o.spmm_csr_bf16(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), np.abs(vals),
                          o.bf16_bits(np.abs(o.bf16_from_bits(h16.numpy()))))
    assert np.all(np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 2.0 ** -19 * mag)     # final rounding + fp32 accumulation (4096-entry rows)
    assert (out.numpy() == o.bf16_bits(ref.astype(np.float32))).mean() > 0.97
    with pytest.raises(GcnxError, match="f in"):
        D.spmm_bf16(ctx, a, ctx.empty((hb.n, 96), np.uint16), None, ctx.empty((hb.n, 96), np.uint16))
    D.spmm_bf16(ctx, D.DeviceCSR.from_host_csr(ctx, np.zeros(1, np.int32), np.zeros(0, np.int32), None, None),
                ctx.empty((0, 64), np.uint16), None, ctx.empty((0, 64), np.uint16))   # empty input: a no-op


@pytest.mark.parametrize("prec", ["bf16", "bf16x3"])
@pytest.mark.parametrize("n", [40000, 33001])
def test_gemm_relu_bit_image_between_forward_and_dx(ctx, n, prec):
    """gcnx_gemm_relu_bits / gcnx_gemm_dx_bits (streaming bf16 kernel): the forward product also writes [out > 0] as a bit
    image, the backward product masks dX with it -- both bit-identical to gcnx_gemm(act = relu) / gcnx_gemm_dx(y_mask) on
    the saved activation, db included; ragged last row block; shapes the kernel does not serve are refused (no launch)."""
    from gcnx import device as D
    rng = np.random.default_rng(n)
    x = ctx.to_device(rng.standard_normal((n, 256), dtype=np.float32))
    w1 = ctx.to_device((rng.standard_normal((256, 256)) / 16).astype(np.float32)); b1 = ctx.to_device(rng.standard_normal(256).astype(np.float32))
    w2 = ctx.to_device((rng.standard_normal((256, 256)) / 16).astype(np.float32))
    dh = ctx.to_device(rng.standard_normal((n, 256), dtype=np.float32))
    y_ref, y_bits = ctx.empty((n, 256)), ctx.empty((n, 256))
    bits = ctx.zeros(16 * n, np.int32)
    D.gemm(ctx, x, w1, b1, y_ref, act="relu", prec=prec)
    assert D.gemm_relu_bits(ctx, x, w1, b1, y_bits, bits, prec=prec)
    assert np.array_equal(y_ref.numpy(), y_bits.numpy())
    assert 0.3 < (y_ref.numpy() > 0).mean() < 0.7
    dx_ref, dx_bits, db_ref, db_bits = ctx.empty((n, 256)), ctx.empty((n, 256)), ctx.empty(256), ctx.empty(256)
    D.gemm_dx(ctx, dh, w2, dx_ref, prec=prec, y_mask=y_ref, db=db_ref)
    D.gemm_dx(ctx, dh, w2, dx_bits, prec=prec, db=db_bits, mask_bits=bits)
    assert np.array_equal(dx_ref.numpy(), dx_bits.numpy())
    # db: the same numbers; the one-plane bit-image form adds them per lane over the whole walk (registers) and combines lanes
    # once, the other forms combine the 16 lanes of a tile per row block -- a different association, not a different sum
    if prec == "bf16x3":
        assert np.array_equal(db_ref.numpy(), db_bits.numpy())
    assert rel_err(db_bits.numpy(), dx_ref.numpy().astype(np.float64).sum(0)) < 2e-6 and rel_err(db_ref.numpy(), db_bits.numpy()) < 2e-6
    assert np.array_equal(dx_bits.numpy() != 0, (y_ref.numpy() > 0) & (dx_bits.numpy() != 0))      # zero wherever the mask is
    # not served: fp32 products, short inputs -- refused without a launch
    assert not D.gemm_relu_bits(ctx, x, w1, b1, y_bits, bits, prec="f32")
    small = ctx.to_device(rng.standard_normal((1000, 256), dtype=np.float32))
    assert not D.gemm_relu_bits(ctx, small, w1, b1, ctx.empty((1000, 256)), bits, prec=prec)


@pytest.mark.parametrize("mode", ["sum", "avg", "max"])
def test_segment_pool_forward_backward(ctx, mode):
    from gcnx import device as D
    from gcnx.device import Segments
    o = O()
    rng = np.random.default_rng(3)
    sizes = [1, 700, 33, 2, 257]
    gp = np.concatenate([[0], np.cumsum(sizes)])
    n, f = int(gp[-1]), 70
    x = rng.standard_normal((n, f), dtype=np.float32)
    x[5:9, 3] = x[5:9, 3].max() + 1.0        # a tie inside graph 1: first maximal row must win
    seg = Segments(ctx, gp)
    pooled = ctx.empty((5, f)); arg = ctx.empty((5, f), np.int32) if mode == "max" else None
    D.segment_pool(ctx, seg, ctx.to_device(x), pooled, mode, arg)
    ref, rarg = o.global_pool_fwd(x.astype(np.float64), gp, mode)
    assert rel_err(pooled.numpy(), ref) < TIGHT
    if mode == "max":
        assert np.array_equal(arg.numpy(), rarg)
    dp = rng.standard_normal((5, f), dtype=np.float32)
    y = rng.standard_normal((n, f), dtype=np.float32)
    dx = ctx.empty((n, f)); db = ctx.empty(f)
    D.segment_pool_bwd(ctx, seg, ctx.to_device(dp), dx, mode, arg)
    rdx = o.global_pool_bwd(dp.astype(np.float64), gp, n, mode, rarg)
    assert rel_err(dx.numpy(), rdx) < TIGHT
    D.segment_pool_bwd(ctx, seg, ctx.to_device(dp), dx, mode, arg, y=ctx.to_device(y), db=db)   # fused mask + db
    rm = rdx * (y > 0)
    assert rel_err(dx.numpy(), rm) < TIGHT and rel_err(db.numpy(), rm.sum(0)) < TIGHT


@pytest.mark.parametrize("mode", ["sum", "avg"])
@pytest.mark.parametrize("shape", ["ecoli128", "ecoli16", "ecoli256", "hub64", "tiny"])
def test_spmm_pool_bwd_fold_equals_pool_bwd_then_spmm(ctx, shape, mode):
    """gcnx_spmm_csr_pool_bwd / gcnx_pool_bwd_colsum against the oracle's unfused chain: pool' -> ReLU mask ->
    A^T (and the column sums of the masked gradient), non-symmetric values so that the transpose matters."""
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR, Segments
    o = O()
    if shape == "hub64":
        hb = synth.power_law_batch(n_graphs=2, graph_size=4096, f=64, seed=5)     # rows split over the workgroup
    elif shape == "tiny":
        hb = synth.ecoli_batch(1, 4, seed=2)
    else:
        hb = synth.ecoli_batch(5, int(shape[5:]), seed=11)
    n, f, b = hb.n, hb.x.shape[1], len(hb.graph_ptr) - 1
    rng = np.random.default_rng(7)
    vals = (rng.random(len(hb.colidx)) + 0.25).astype(np.float32)
    y = np.maximum(rng.standard_normal((n, f), dtype=np.float32), 0)              # a saved ReLU output (about half zeros)
    dp = rng.standard_normal((b, f), dtype=np.float32)
    dz = o.global_pool_bwd(dp.astype(np.float64), hb.graph_ptr, n, mode, None) * (y > 0)
    ref = o.spmm_csr_T(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals.astype(np.float64), dz)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr, symmetric=False)
    seg = Segments(ctx, hb.graph_ptr)
    out = ctx.empty((n, f)); db = ctx.empty(f)
    D.spmm_pool_bwd(ctx, a.transpose(), ctx.to_device(y), seg, ctx.to_device(dp), out, mode)
    assert rel_err(out.numpy(), ref) < TIGHT
    D.pool_bwd_colsum(ctx, seg, ctx.to_device(dp), ctx.to_device(y), db, mode)
    assert rel_err(db.numpy(), dz.sum(0)) < TIGHT
    # unweighted operator (GeneralConv aggregation)
    au = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    D.spmm_pool_bwd(ctx, au.transpose(), ctx.to_device(y), seg, ctx.to_device(dp), out, mode)
    assert rel_err(out.numpy(), o.spmm_csr_T(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None, dz)) < TIGHT


@pytest.mark.parametrize("mode", ["sum", "avg"])
def test_spmm_pool_bwd_fold_on_the_tile_kernels(ctx, mode):
    """The folded backward aggregation with a tile plan (throughput regime: >= 128 graphs): both tile tiers mask their
    LDS tile in place, graphs taller than a tile go through the rows kernel's chunk list; against the oracle's unfused
    chain, non-symmetric values."""
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR, Segments
    o = O()
    hb = synth.block_diag_batch(120_000, 1_200_000, 64, seed=4)
    sizes = np.diff(hb.graph_ptr)
    assert len(sizes) >= 128 and (sizes <= 604).any() and ((sizes > 604) & (sizes <= 1236)).any() and (sizes > 1236).any()
    n, f, b = hb.n, 64, len(sizes)
    rng = np.random.default_rng(8)
    vals = (rng.random(len(hb.colidx)) + 0.25).astype(np.float32)
    y = np.maximum(rng.standard_normal((n, f), dtype=np.float32), 0)
    dp = rng.standard_normal((b, f), dtype=np.float32)
    dz = o.global_pool_bwd(dp.astype(np.float64), hb.graph_ptr, n, mode, None) * (y > 0)
    ref = o.spmm_csr_T(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals.astype(np.float64), dz)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr, symmetric=False)
    at = a.transpose()
    assert at.plan is not None
    seg = Segments(ctx, hb.graph_ptr)
    out = ctx.empty((n, f)); dy, ddp = ctx.to_device(y), ctx.to_device(dp)
    ctx.set_tuning("spmm_kernel", "tile")      # (200 graphs x 2 slabs are fewer units than the library asks for before it picks the tiles)
    D.spmm_pool_bwd(ctx, at, dy, seg, ddp, out, mode)
    assert rel_err(out.numpy(), ref) < TIGHT
    tiled = out.numpy().copy()
    # ... and from the bit image the forward launch of the pooled layer writes (gcnx_spmm_csr_relu_bits): y2 = relu(A h + b)
    hsrc = rng.standard_normal((n, f), dtype=np.float32); bias = rng.standard_normal(f).astype(np.float32)
    y2 = ctx.empty((n, f)); bits = ctx.zeros((f // 32) * n, np.int32)
    assert D.spmm_relu_bits(ctx, a, ctx.to_device(hsrc), ctx.to_device(bias), y2, bits)
    y2h = y2.numpy()
    ref_y2 = np.maximum(o.spmm_csr(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals.astype(np.float64),
                                   hsrc.astype(np.float64)) + bias, 0)
    assert rel_err(y2h, ref_y2) < TIGHT
    img = bits.numpy().view(np.uint32).reshape(f // 32, n)                  # slab-major: one word per (slab, row)
    tile_rows = np.concatenate([np.arange(hb.graph_ptr[g], hb.graph_ptr[g + 1]) for g in range(b) if sizes[g] <= 1236])
    got = ((img[:, tile_rows, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)          # [slab, row, bit]
    want = (y2h[tile_rows] > 0).reshape(len(tile_rows), f // 32, 32).transpose(1, 0, 2)
    assert np.array_equal(got, want)
    dz2 = o.global_pool_bwd(dp.astype(np.float64), hb.graph_ptr, n, mode, None) * (y2h > 0)
    ref2 = o.spmm_csr_T(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals.astype(np.float64), dz2)
    D.spmm_pool_bwd(ctx, at, y2, seg, ddp, out, mode, y_bits=bits)
    assert rel_err(out.numpy(), ref2) < TIGHT
    from_bits = out.numpy().copy()
    D.spmm_pool_bwd(ctx, at, y2, seg, ddp, out, mode)                       # the fp32-source fold gives the same bits of dH
    assert np.array_equal(out.numpy(), from_bits)
    ctx.set_tuning("spmm_kernel", "rows")                                   # the same call on the row gather
    try:
        D.spmm_pool_bwd(ctx, at, dy, seg, ddp, out, mode)
        assert not D.spmm_relu_bits(ctx, a, ctx.to_device(hsrc), ctx.to_device(bias), y2, bits)   # no tiles, no bit image
    finally:
        ctx.set_tuning("spmm_kernel", "auto")
    assert rel_err(out.numpy(), ref) < TIGHT and rel_err(out.numpy(), tiled) < TIGHT


def test_spmm_pool_bwd_fold_argument_errors(ctx):
    from gcnx import device as D, synth
    from gcnx.device import Segments
    from gcnx._lib import GcnxError
    hb = synth.ecoli_batch(2, 10, seed=1)
    a, _ = _csr(ctx, hb, True)
    seg = Segments(ctx, hb.graph_ptr)
    y = ctx.to_device(hb.x); dp = ctx.zeros((2, 10)); out = ctx.empty((hb.n, 10)); db = ctx.empty(10)
    with pytest.raises(GcnxError, match="multiples of 4"):
        D.spmm_pool_bwd(ctx, a, y, seg, dp, out)                  # f = 10: the caller takes the unfused pair
    with pytest.raises(GcnxError, match="no folded form"):
        D.pool_bwd_colsum(ctx, seg, dp, y, db, "max")
    D.pool_bwd_colsum(ctx, seg, dp, y, db)                        # scalar-width columns are fine for the column sums
    assert np.array_equal(db.numpy(), np.zeros(10, np.float32))


@pytest.mark.parametrize("cce", ["logits", "probs"])
def test_softmax_cce_matches_keras_semantics(ctx, cce):
    """Both code paths of keras.backend.categorical_crossentropy (include/gcnx.h gcnx_cce_mode): "logits" = what
    train_step runs under tf.function, "probs" = the eager renormalise-and-clip form; rows 0/1 are saturated, where
    the two differ (loss 80 vs 16.1 for the mislabelled row, gradient -1/B vs 0)."""
    from gcnx import device as D
    o = O()
    rng = np.random.default_rng(4)
    b, c = 1000, 2
    logits = (4 * rng.standard_normal((b, c))).astype(np.float32)
    logits[0] = [40.0, -40.0]; logits[1] = [-40.0, 40.0]            # saturated: exercises the clip
    y = np.eye(c, dtype=np.float32)[rng.integers(0, c, b)]; y[0] = [1, 0]; y[1] = [1, 0]
    probs = ctx.empty((b, c)); la = ctx.zeros(2); dl = ctx.empty((b, c))
    D.softmax_cce(ctx, ctx.to_device(logits), ctx.to_device(y), probs, la, dl, denom=b, cce=cce)
    z64, y64 = logits.astype(np.float64), y.astype(np.float64)
    p = o.softmax(z64)
    rl, rdl = o.cce(y64, z64, p, None, cce)
    assert rel_err(probs.numpy(), p) < TIGHT
    loss, hits = la.numpy()
    assert abs(loss - rl) < 1e-5 * max(1, loss)
    assert hits == round(o.categorical_accuracy(y, p) * b)
    assert rel_err(dl.numpy(), rdl) < TIGHT
    assert (dl.numpy()[1, 0] == 0.0) == (cce == "probs")            # the clipped row gets no gradient in the eager form only
    # shard semantics: denom = global batch
    la.fill_zero()
    D.softmax_cce(ctx, ctx.to_device(logits[:10]), ctx.to_device(y[:10]), ctx.empty((10, c)), la, None, denom=40, cce=cce)
    assert abs(la.numpy()[0] - o.cce(y64[:10], z64[:10], p[:10], 40, cce)[0]) < 1e-5


@pytest.mark.parametrize("cce", ["logits", "probs"])
@pytest.mark.parametrize("b,h,c", [(32, 128, 2), (1, 16, 1), (1667, 256, 2), (200, 40, 7), (40, 2000, 3)])
def test_dense_softmax_cce_head(ctx, b, h, c, cce):
    """gcnx_dense_softmax_cce (one launch) == Dense + softmax + clipped CCE + accuracy + the head gradients of the
    oracle; several workgroups (b > 64) reduce dW/db/loss in a fixed order: two launches agree bitwise."""
    from gcnx import device as D
    o = O()
    rng = np.random.default_rng(11)
    pooled = (3 * rng.standard_normal((b, h))).astype(np.float32)
    w = (rng.standard_normal((h, c)) / np.sqrt(h)).astype(np.float32)
    bias = rng.standard_normal(c).astype(np.float32)
    y = np.eye(c, dtype=np.float32)[rng.integers(0, c, b)]
    if c == 2 and b >= 2:
        pooled[0] = 0; pooled[0, 0] = 400.0; w[0] = [1.0, -1.0]      # saturated graphs: exercise the clip / the no-clip form
        pooled[1] = 0; pooled[1, 0] = -30.0; y[0] = [0, 1]; y[1] = [1, 0]
    denom = float(b + 3)                                             # a "global batch" larger than this shard
    dp, dw_, db_ = ctx.empty((b, h)), ctx.empty((h, c)), ctx.empty(c)
    probs, la = ctx.empty((b, c)), ctx.to_device(np.array([7.0, 7.0], np.float32))   # overwritten, not accumulated
    args = (ctx, ctx.to_device(pooled), ctx.to_device(w), ctx.to_device(bias), ctx.to_device(y), probs, la, denom)
    D.dense_softmax_cce(*args, dw=dw_, db=db_, dpooled=dp, cce=cce)
    P, W, Y = pooled.astype(np.float64), w.astype(np.float64), y.astype(np.float64)
    z = P @ W + bias.astype(np.float64)
    p = o.softmax(z)
    rl, dl = o.cce(Y, z, p, denom, cce)
    assert rel_err(probs.numpy(), p) < TIGHT
    loss, hits = la.numpy()
    assert abs(loss - rl) < 2e-5 * max(1.0, abs(loss))
    assert hits == round(o.categorical_accuracy(y, p) * b)
    assert rel_err(dw_.numpy(), P.T @ dl) < TIGHT and rel_err(db_.numpy(), dl.sum(0)) < TIGHT
    assert rel_err(dp.numpy(), dl @ W.T) < TIGHT
    first = (dw_.numpy().copy(), db_.numpy().copy(), la.numpy().copy())
    D.dense_softmax_cce(*args, dw=dw_, db=db_, dpooled=dp, cce=cce)
    assert np.array_equal(first[0], dw_.numpy()) and np.array_equal(first[1], db_.numpy()) and np.array_equal(first[2], la.numpy())
    # loss only, and probabilities only
    la2 = ctx.zeros(2)
    D.dense_softmax_cce(ctx, args[1], args[2], args[3], args[4], probs, la2, denom, cce=cce)
    assert np.array_equal(la2.numpy(), first[2])
    pr2 = ctx.empty((b, c))
    D.dense_softmax_cce(ctx, args[1], args[2], args[3], None, pr2)
    assert np.array_equal(pr2.numpy(), probs.numpy())


@pytest.mark.parametrize("b,h,c,mode", [(32, 128, 2, "sum"), (5, 16, 3, "avg"), (100, 64, 2, "sum"), (100, 64, 2, "avg"),
                                        (700, 64, 2, "sum"), (700, 64, 2, "avg"), (1500, 32, 3, "avg"), (6, 32, 2, "max"),
                                        (3, 600, 2, "sum")])
def test_pool_dense_softmax_cce_equals_the_two_calls(ctx, b, h, c, mode):
    """gcnx_pool_dense_softmax_cce: split pools (few graphs, sum/avg) are combined by the head itself; one or several
    head workgroups (partials and dW slabs share the workspace), MAX / wide operands take the two calls, many graphs
    the two calls with db_relu computed from the pool's own count of positive entries.
    Every output must equal gcnx_segment_pool + gcnx_dense_softmax_cce up to the pool's fp32 summation order (the
    fused form slices the rows differently), and two runs of the fused call must agree bit for bit."""
    from gcnx import device as D
    from gcnx.device import Segments
    rng = np.random.default_rng(b + h)
    sizes = rng.integers(1, 90, b); sizes[0] = 1
    gp = np.concatenate([[0], np.cumsum(sizes)])
    n = int(gp[-1])
    x = ctx.to_device(rng.standard_normal((n, h), dtype=np.float32))
    w = ctx.to_device((rng.standard_normal((h, c)) / np.sqrt(h)).astype(np.float32))
    bias = ctx.to_device(rng.standard_normal(c).astype(np.float32))
    y = ctx.to_device(np.eye(c, dtype=np.float32)[rng.integers(0, c, b)])
    seg = Segments(ctx, gp)
    out = {}
    for fused in (False, True, "again"):
        pooled, probs, la = ctx.zeros((b, h)), ctx.empty((b, c)), ctx.zeros(2)
        dw_, db_, dp = ctx.empty((h, c)), ctx.empty(c), ctx.empty((b, h))
        dbr = ctx.zeros(h)
        arg = ctx.empty((b, h), np.int32) if mode == "max" else None
        if fused:
            D.pool_dense_softmax_cce(ctx, seg, x, pooled, w, bias, y, probs, la, float(b), dw=dw_, db=db_, dpooled=dp,
                                     mode=mode, argmax=arg, db_relu=None if mode == "max" else dbr)
        else:
            D.segment_pool(ctx, seg, x, pooled, mode, arg)
            D.dense_softmax_cce(ctx, pooled, w, bias, y, probs, la, float(b), dw=dw_, db=db_, dpooled=dp)
            if mode != "max":
                D.pool_bwd_colsum(ctx, seg, dp, x, dbr, mode)
        out[fused] = [a.numpy() for a in (pooled, probs, la, dw_, db_, dp, dbr)]
    for a0, a1, a2 in zip(out[False], out[True], out["again"]):
        assert rel_err(a1, a0) < TIGHT and np.array_equal(a1, a2)
    assert out[True][2][1] == out[False][2][1]                       # hit count
    pr = ctx.empty((b, c)); pooled = ctx.zeros((b, h))
    D.pool_dense_softmax_cce(ctx, seg, x, pooled, w, bias, None, pr, mode=mode,
                             argmax=ctx.empty((b, h), np.int32) if mode == "max" else None)    # inference surface
    assert np.array_equal(pr.numpy(), out[True][1]) and np.array_equal(pooled.numpy(), out[True][0])


@pytest.mark.parametrize("act", [None, "relu", "prelu"])
def test_act_bias_grad(ctx, act):
    from gcnx import device as D
    o = O()
    rng = np.random.default_rng(6)
    n, f = 5000, 96
    dy = rng.standard_normal((n, f), dtype=np.float32); y = rng.standard_normal((n, f), dtype=np.float32)
    al = rng.random(f).astype(np.float32)
    dz = ctx.empty((n, f)); db = ctx.empty(f); dal = ctx.empty(f)
    D.act_bias_grad(ctx, ctx.to_device(dy), ctx.to_device(y), dz, act, db=db,
                    alpha=ctx.to_device(al) if act == "prelu" else None, dalpha=dal if act == "prelu" else None)
    ref = o.act_bwd(dy.astype(np.float64), y.astype(np.float64), act, al.astype(np.float64))
    assert rel_err(dz.numpy(), ref) < TIGHT and rel_err(db.numpy(), ref.sum(0)) < TIGHT
    if act == "prelu":
        assert rel_err(dal.numpy(), (dy.astype(np.float64) * np.minimum(y, 0)).sum(0)) < TIGHT


@pytest.mark.parametrize("n,f", [(70001, 256), (65536, 64), (131075, 16), (70001, 48)])
def test_colsum_tall(ctx, n, f):
    """Plain column sums of a tall matrix (db of a million-row layer): the whole-row kernel (f a power of two) and the
    64-column-tile kernel (f = 48) against float64 sums."""
    from gcnx import device as D
    rng = np.random.default_rng(n + f)
    x = rng.standard_normal((n, f), dtype=np.float32)
    dx = ctx.to_device(x); db = ctx.empty(f)
    D.act_bias_grad(ctx, dx, None, dx, None, db=db)
    ref = x.astype(np.float64).sum(0)
    assert np.abs(db.numpy() - ref).max() < 1e-5 * np.abs(x).sum(0).max()
    first = db.numpy().copy()
    D.act_bias_grad(ctx, dx, None, dx, None, db=db)
    assert np.array_equal(db.numpy(), first)          # fixed-order reduction: run-to-run identical


def test_graph_prep_coo_to_csr_norm_transpose(ctx):
    from gcnx import _lib, synth
    from gcnx.device import DeviceCSR
    o = O()
    hb = synth.ecoli_batch(3, 4, seed=8)
    idx = hb.indices()
    a = DeviceCSR.from_coo(ctx, idx, np.ones(len(idx)), hb.n, graph_ptr=hb.graph_ptr)
    assert np.array_equal(a.rowptr.numpy(), hb.rowptr) and np.array_equal(a.colidx.numpy()[:hb.nnz], hb.colidx)
    an = a.unweighted().gcn_norm("spektral")
    ref = o.gcn_filter_csr(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None, "spektral")
    assert rel_err(an.vals.numpy()[:hb.nnz], ref) < 1e-6
    ap = a.unweighted().gcn_norm("pyg")
    assert rel_err(ap.vals.numpy()[:hb.nnz], o.gcn_filter_csr(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None, "pyg")) < 1e-6
    # empty rows in the COO (isolated nodes) and an empty matrix
    idx2 = np.array([[1, 1], [1, 3], [4, 0]], np.int64)
    a2 = DeviceCSR.from_coo(ctx, idx2, None, 6)
    assert np.array_equal(a2.rowptr.numpy(), [0, 0, 2, 2, 2, 3, 3])
    a0 = DeviceCSR.from_coo(ctx, np.zeros((0, 2), np.int64), None, 3)
    assert np.array_equal(a0.rowptr.numpy(), [0, 0, 0, 0])
    # unsorted rows / out-of-range indices are rejected (GCNX_ERR_DATA), not mis-converted
    with pytest.raises(_lib.GcnxError) as e:
        DeviceCSR.from_coo(ctx, idx2[::-1].copy(), None, 6)
    assert e.value.code == 6
    with pytest.raises(_lib.GcnxError):
        DeviceCSR.from_coo(ctx, np.array([[0, 9]], np.int64), None, 6)
    with pytest.raises(_lib.GcnxError, match="diagonal"):
        a2.gcn_norm()
    # transpose of an asymmetric weighted matrix
    rng = np.random.default_rng(0)
    rows = np.sort(rng.integers(0, 50, 400)); cols = rng.integers(0, 50, 400)
    key = np.unique(rows * 50 + cols); rows, cols = key // 50, key % 50
    vals = rng.random(len(key)).astype(np.float32)
    rp = np.zeros(51, np.int32); np.cumsum(np.bincount(rows, minlength=50), out=rp[1:])
    at = DeviceCSR.from_host_csr(ctx, rp, cols.astype(np.int32), vals, symmetric=False).transpose()
    trp, tci, tv = o.csr_transpose(rp.astype(np.int64), cols, vals)
    assert np.array_equal(at.rowptr.numpy(), trp) and np.array_equal(at.colidx.numpy(), tci) and np.array_equal(at.vals.numpy(), tv)


def test_h2d_async_small_copies_in_stream_order(ctx):
    """gcnx_h2d_async: the bytes are taken before the call returns (the source is overwritten right after), the copies
    land in stream order, more of them than the staging ring has slots, and a copy larger than a slot takes gcnx_h2d."""
    rng = np.random.default_rng(4)
    dst = [ctx.empty(100, np.int32) for _ in range(80)]
    src = np.empty(100, np.int32)
    want = []
    for d in dst:
        src[:] = rng.integers(0, 1 << 30, 100)
        want.append(src.copy())
        d.copy_from_host(src, wait=False)
        src[:] = -1
    for d, w in zip(dst, want):
        assert np.array_equal(d.numpy(), w)
    big = rng.integers(0, 1 << 30, 10000).astype(np.int32)
    dbig = ctx.empty(10000, np.int32)
    dbig.copy_from_host(big, wait=False)
    assert np.array_equal(dbig.numpy(), big)


def test_sgd_and_graph_capture_replay(ctx):
    from gcnx import device as D
    rng = np.random.default_rng(2)
    p = rng.standard_normal(100003).astype(np.float32); g = rng.standard_normal(100003).astype(np.float32)
    dp, dg = ctx.to_device(p), ctx.to_device(g)
    D.sgd(ctx, dp, dg, 0.02)
    # the kernel contracts p - lr*g into one fma (single rounding): compare in fp64, 1 ulp slack
    ref = p.astype(np.float64) - 0.02 * g.astype(np.float64)
    assert np.allclose(dp.numpy(), ref, rtol=0, atol=5e-7)
    once = dp.numpy()
    graph = ctx.capture(lambda: D.sgd(ctx, dp, dg, 0.02))      # captured, not yet executed
    assert np.array_equal(dp.numpy(), once)
    graph.launch(); graph.launch(); ctx.sync()
    assert np.allclose(dp.numpy(), p.astype(np.float64) - 0.06 * g.astype(np.float64), rtol=0, atol=2e-6)
    graph.destroy()
    t0, t1 = ctx.event().record(), None
    D.sgd(ctx, dp, dg, 0.0)
    t1 = ctx.event().record()
    assert t1.elapsed_ms_since(t0) >= 0.0
    info = ctx.info()
    assert info["arch"].startswith("gfx950") and info["cus"] >= 200


# ---- GCNConv as one launch (csrc/fused.hip) -------------------------------------------------------------------------

@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("fi,fo", [(128, 128), (64, 128), (32, 16), (128, 48), (64, 64)])
@pytest.mark.parametrize("weighted", [False, True])
def test_gcn_conv_fused_forward(ctx, fi, fo, weighted, prec):
    """gcnx_gcn_conv_fwd = relu((A X) W + b) and S = A X against the float64 oracle chain dense -> aggregation
    (GCNConv.call order A (X W)); ragged last tile (N % 32 != 0), rows with more than 4 entries per trip."""
    from gcnx import device as D, synth
    o = O()
    hb = synth.ecoli_batch(3, fi, seed=fi + fo)
    assert hb.n % 32 != 0
    a, vals = _csr(ctx, hb, weighted)
    rng = np.random.default_rng(fi * fo)
    w = (rng.standard_normal((fi, fo)) / np.sqrt(fi)).astype(np.float32)
    bias = rng.standard_normal(fo).astype(np.float32)
    assert D.gcn_conv_fused_ok(ctx, hb.n, fi, fo)
    out = ctx.empty((hb.n, fo)); s = ctx.empty((hb.n, fi)); wt = ctx.zeros((fo, fi))
    D.gcn_conv_fwd(ctx, a, ctx.to_device(hb.x), ctx.to_device(w), ctx.to_device(bias), out, act="relu", s=s, wt=wt, prec=prec)
    assert np.array_equal(wt.numpy(), w.T)                                # the transposed weight by-product
    ref = _ref_spmm(hb, vals, hb.x.astype(np.float64) @ w.astype(np.float64), bias, True)
    tol = TIGHT if prec == "f32" else 5e-5                                # split-bf16 products: 2^-17 per operand
    assert rel_err(out.numpy(), ref) < tol
    assert rel_err(s.numpy(), _ref_spmm(hb, vals, hb.x)) < TIGHT            # the gather is fp32 either way
    D.gcn_conv_fwd(ctx, a, ctx.to_device(hb.x), ctx.to_device(w), None, out, act=None, prec=prec)   # inference surface: no S
    assert rel_err(out.numpy(), _ref_spmm(hb, vals, hb.x.astype(np.float64) @ w.astype(np.float64))) < tol


def test_gcn_conv_fused_long_rows_and_refusals(ctx):
    """A hub row with 4096 entries overflows the tile's LDS staging (1024 entries): the overflow path; widths the
    kernel does not serve are refused with GCNX_ERR_UNSUPPORTED."""
    from gcnx import device as D, synth
    from gcnx._lib import GcnxError
    hb = synth.power_law_batch(n_graphs=1, graph_size=8192, f=64, seed=3)
    a, vals = _csr(ctx, hb, True)
    rng = np.random.default_rng(1)
    w = (rng.standard_normal((64, 32)) / 8).astype(np.float32)
    out = ctx.empty((hb.n, 32))
    D.gcn_conv_fwd(ctx, a, ctx.to_device(hb.x), ctx.to_device(w), None, out, act="relu")
    assert rel_err(out.numpy(), _ref_spmm(hb, vals, hb.x.astype(np.float64) @ w.astype(np.float64), None, True)) < TIGHT
    assert not D.gcn_conv_fused_ok(ctx, hb.n, 96, 32) and not D.gcn_conv_fused_ok(ctx, hb.n, 64, 256)
    with pytest.raises(GcnxError, match="gcnx_gemm \\+ gcnx_spmm_csr"):
        D.gcn_conv_fwd(ctx, a, ctx.to_device(hb.x), ctx.zeros((64, 256)), None, ctx.empty((hb.n, 256)))


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("f1,f2", [(128, 128), (64, 128), (128, 32)])
@pytest.mark.parametrize("mode", ["sum", "avg"])
def test_gcn_conv_fused_backward_and_dw2(ctx, f1, f2, mode, prec):
    """gcnx_gcn_conv_bwd_pool + gcnx_gemm_dw2 against the oracle's unfused chain: pool' -> ReLU' -> A^T -> W2^T -> ReLU',
    column sums, the two weight gradients and the SGD step; non-symmetric values so that the transpose matters; the db1
    reduction both immediate and pending."""
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR, Segments
    o = O()
    hb = synth.ecoli_batch(4, f2, seed=f1 + f2)
    n, b = hb.n, len(hb.graph_ptr) - 1
    rng = np.random.default_rng(17)
    vals = (rng.random(len(hb.colidx)) + 0.25).astype(np.float32)
    y2 = np.maximum(rng.standard_normal((n, f2), dtype=np.float32), 0)
    y1 = np.maximum(rng.standard_normal((n, f1), dtype=np.float32), 0)
    dp = rng.standard_normal((b, f2), dtype=np.float32)
    w2 = (rng.standard_normal((f1, f2)) / np.sqrt(f1)).astype(np.float32)
    s1 = rng.standard_normal((n, 32), dtype=np.float32); s2 = rng.standard_normal((n, f1), dtype=np.float32)
    rdz2 = o.global_pool_bwd(dp.astype(np.float64), hb.graph_ptr, n, mode, None) * (y2 > 0)
    rdh = o.spmm_csr_T(hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), vals.astype(np.float64), rdz2)
    rdz1 = (rdh @ w2.astype(np.float64).T) * (y1 > 0)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr, symmetric=False)
    seg = Segments(ctx, hb.graph_ptr)
    dz2 = ctx.empty((n, f2)); dz1 = ctx.empty((n, f1)); db1 = ctx.empty(f1)
    args = (ctx, a.transpose(), ctx.to_device(y2), seg, ctx.to_device(dp), ctx.to_device(w2), ctx.to_device(y1), dz2, dz1)
    tol = TIGHT if prec == "f32" else 5e-5
    pend = D.gcn_conv_bwd_pool(*args, db1=db1, mode=mode, prec=prec)
    assert not pend.colpart                                              # no scratch: db1 is final
    assert rel_err(dz2.numpy(), rdz2) < TIGHT and rel_err(dz1.numpy(), rdz1) < tol
    assert rel_err(db1.numpy(), rdz1.sum(0)) < tol
    first = (dz2.numpy().copy(), dz1.numpy().copy(), db1.numpy().copy())
    D.gcn_conv_bwd_pool(*args, db1=db1, mode=mode, w2t=ctx.to_device(np.ascontiguousarray(w2.T)), prec=prec)   # weight operand pre-transposed
    assert all(np.array_equal(x, y) for x, y in zip(first, (dz2.numpy(), dz1.numpy(), db1.numpy())))
    # pending form + both weight gradients + SGD in the flat buffers
    n_par = 32 * f1 + f1 * f2 + f1
    params = ctx.to_device(rng.standard_normal(n_par, dtype=np.float32)); grads = ctx.zeros(n_par)
    p0 = params.numpy().copy()
    gw1, gw2, gb1 = grads.flat(0, 32 * f1, (32, f1)), grads.flat(32 * f1, f1 * f2, (f1, f2)), grads.flat(32 * f1 + f1 * f2, f1)
    scratch = ctx.empty(D.gcn_conv_bwd_scratch_floats(ctx, n, f1))
    pend = D.gcn_conv_bwd_pool(*args, db1=gb1, mode=mode, scratch=scratch, prec=prec)
    assert pend.colpart
    D.gemm_dw2(ctx, ctx.to_device(s1), dz1, gw1, ctx.to_device(s2), dz2, gw2, params=params, grads=grads, lr=0.05, pending=pend)
    rg = np.concatenate([(s1.astype(np.float64).T @ rdz1).ravel(), (s2.astype(np.float64).T @ rdz2).ravel(), rdz1.sum(0)])
    assert rel_err(grads.numpy(), rg) < tol
    assert rel_err(params.numpy(), p0 - 0.05 * rg) < tol
    # gradients only (the multi-GPU path: all-reduce first, update later)
    grads.fill_zero()
    pend = D.gcn_conv_bwd_pool(*args, db1=gb1, mode=mode, scratch=scratch, prec=prec)
    D.gemm_dw2(ctx, ctx.to_device(s1), dz1, gw1, ctx.to_device(s2), dz2, gw2, grads=grads, pending=pend)
    assert rel_err(grads.numpy(), rg) < tol


@pytest.mark.parametrize("cce", ["logits", "probs"])
@pytest.mark.parametrize("mode", ["sum", "avg"])
@pytest.mark.parametrize("shape,h,c", [("ecoli", 128, 2), ("ecoli", 32, 1), ("ragged", 64, 2), ("many", 32, 2), ("tall", 128, 2),
                                       ("saturated", 64, 2)])
def test_head_inside_the_backward_launches(ctx, shape, h, c, mode, cce):
    """gcnx_head_args: the forward launch leaves the pool's per-tile partial sums (gcnx_gcn_conv_fwd_pool) ->
    gcnx_gcn_conv_bwd_pool(head) adds them up and evaluates dPooled per graph itself -> gcnx_gemm_dw2(leaf) produces the
    head's outputs, against the sequence pool + head launch -> backward -> dw2 with the same operands.  "ragged":
    single-node graphs, so 32-row tiles span up to 13 graphs (several rounds of the in-kernel head); "many": more graphs
    than one head workgroup holds (the leaf falls back to a launch of its own); "tall": graphs of 72 and 35 tiles (more
    partial rows than one pass of the strided loads covers); "saturated": logits in the hundreds (the clip branch of the
    eager CCE form inside the kernels: no gradient); "ecoli" at h = 128, c = 2 is the shape the merged launch serves.
    Two runs agree bit for bit."""
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR, Segments
    rng = np.random.default_rng(h + c)
    if shape in ("ecoli", "saturated"):
        hb = synth.ecoli_batch(4, h, seed=h)
        rowptr, colidx, gp = hb.rowptr, hb.colidx, hb.graph_ptr
    else:
        sizes = (np.array([1, 1, 1, 2, 1, 50, 1, 1, 1, 1, 3, 1, 70, 1, 1], np.int64) if shape == "ragged"
                 else np.array([5, 2300, 31, 1100], np.int64) if shape == "tall"      # 72 and 35 tiles: more than one pass of
                 else rng.integers(1, 40, 90))                                       # the eight waves' strided partial loads
        gp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        rows, cols = [], []
        for g in range(len(sizes)):                                     # self-loops and a chain inside every graph
            for i in range(gp[g], gp[g + 1]):
                rows.append(i); cols.append(i)
                if i + 1 < gp[g + 1]:
                    rows += [i, i + 1]; cols += [i + 1, i]
        order = np.lexsort((cols, rows))
        rows, cols = np.asarray(rows)[order], np.asarray(cols)[order]
        rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=int(gp[-1])))]).astype(np.int32)
        colidx = cols.astype(np.int32)
    n, b, f1 = int(gp[-1]), len(gp) - 1, 64
    vals = (rng.random(len(colidx)) + 0.25).astype(np.float32)
    a = DeviceCSR.from_host_csr(ctx, rowptr, colidx, vals, gp, symmetric=False)
    seg = Segments(ctx, gp)
    y1 = ctx.to_device(np.maximum(rng.standard_normal((n, f1), dtype=np.float32), 0))
    w2 = ctx.to_device((rng.standard_normal((f1, h)) / np.sqrt(f1)).astype(np.float32))
    # the pooled layer's forward launch leaves the pool's per-tile partial sums and positive counts: row t + g = the rows
    # of graph g inside tile t
    y2, tp, tc = ctx.empty((n, h)), ctx.zeros((D.pool_tile_rows(n, b), h)), ctx.zeros((D.pool_tile_rows(n, b), h))
    D.gcn_conv_fwd(ctx, a, y1, w2, ctx.to_device(rng.standard_normal(h).astype(np.float32)), y2, act="relu", pool=(seg, tp, tc))
    y2h, tph, tch = y2.numpy(), tp.numpy(), tc.numpy()
    assert (y2h > 0).mean() > 0.2
    for g in range(b):
        tiles = range(gp[g] // 32, (gp[g + 1] - 1) // 32 + 1)
        rows = y2h[gp[g]:gp[g + 1]]
        assert rel_err(sum(tph[t + g].astype(np.float64) for t in tiles), rows.sum(0, dtype=np.float64)) < TIGHT
        assert np.array_equal(sum(tch[t + g] for t in tiles), (rows > 0).sum(0).astype(np.float32))
        for t in tiles:                                                    # and each partial row by itself
            lo, hi = max(gp[g], 32 * t), min(gp[g + 1], 32 * t + 32)
            assert rel_err(tph[t + g], y2h[lo:hi].sum(0, dtype=np.float64)) < TIGHT
    s1 = ctx.to_device(rng.standard_normal((n, 32), dtype=np.float32)); s2 = ctx.to_device(rng.standard_normal((n, f1), dtype=np.float32))
    scale = np.sqrt(h) * (n / b if mode == "sum" else 1.0)              # logits of order 1: no saturated softmax, gradients flow
    if shape == "saturated":                                            # |logits| in the hundreds: probabilities of exactly 0 / 1 --
        scale /= 300.0                                                  # the eager form clips (no gradient), the logits form does not
    w3 = ctx.to_device((rng.standard_normal((h, c)) / scale).astype(np.float32))
    b3 = ctx.to_device(rng.standard_normal(c).astype(np.float32))
    y = ctx.to_device(np.eye(c, dtype=np.float32)[rng.integers(0, c, b)])
    denom = float(b + 1)
    # flat gradient buffer: dW1 | dW2 | db1 | dW3 | db3 | db2   (everything the two launches write)
    offs = [int(v) for v in np.cumsum([0, 32 * f1, f1 * h, f1, h * c, c, h])]
    names = ("pooled", "probs", "la", "dp", "dz2", "dz1", "grads", "params")
    out = {}
    for how in ("early", "late", "late again", "late grads only"):
        params = ctx.to_device(np.linspace(-1, 1, offs[-1] + 3, dtype=np.float32)); grads = ctx.zeros(offs[-1] + 3)
        gw1, gw2, gb1 = grads.flat(offs[0], 32 * f1, (32, f1)), grads.flat(offs[1], f1 * h, (f1, h)), grads.flat(offs[2], f1)
        gw3, gb3, gb2 = grads.flat(offs[3], h * c, (h, c)), grads.flat(offs[4], c), grads.flat(offs[5], h)
        pooled, probs, la, dp = ctx.zeros((b, h)), ctx.zeros((b, c)), ctx.zeros(2), ctx.zeros((b, h))
        dz2, dz1 = ctx.empty((n, h)), ctx.empty((n, f1))
        scratch = ctx.empty(D.gcn_conv_bwd_scratch_floats(ctx, n, f1))
        sgd = dict(grads=grads) if how == "late grads only" else dict(params=params, grads=grads, lr=0.05)
        if how == "early":
            D.pool_dense_softmax_cce(ctx, seg, y2, pooled, w3, b3, y, probs, la, denom, dw=gw3, db=gb3, dpooled=dp, mode=mode,
                                     db_relu=gb2, cce=cce)
            pend = D.gcn_conv_bwd_pool(ctx, a.transpose(), y2, seg, dp, w2, y1, dz2, dz1, db1=gb1, mode=mode, scratch=scratch)
            D.gemm_dw2(ctx, s1, dz1, gw1, s2, dz2, gw2, pending=pend, **sgd)
        else:
            ha = D.head_args(seg, tp, tc, ctx.empty((b, h)), ctx.empty((b, h)), w3, b3, y, denom, probs, la, gw3, gb3, gb2, pooled, dp,
                             mode=mode, cce=cce)
            pend = D.gcn_conv_bwd_pool(ctx, a.transpose(), y2, seg, None, w2, y1, dz2, dz1, db1=gb1, mode=mode, scratch=scratch, head=ha)
            D.gemm_dw2(ctx, s1, dz1, gw1, s2, dz2, gw2, pending=pend, leaf=ha, **sgd)
        out[how] = dict(zip(names, [t.numpy().copy() for t in (pooled, probs, la, dp, dz2, dz1, grads, params)]))
    for k in names:
        assert rel_err(out["late"][k], out["early"][k]) < TIGHT, k
        assert np.array_equal(out["late"][k], out["late again"][k]), k
        if k != "params":
            assert np.array_equal(out["late"][k], out["late grads only"][k]), k
    assert out["late"]["la"][1] == out["early"]["la"][1]                # hit count
    if shape != "saturated":
        assert np.any(out["late"]["grads"][offs[5]:offs[6]] != 0) or c == 1  # db2 was produced (c == 1: all gradients vanish)
    else:
        p_ = out["late"]["probs"]
        assert np.all((p_ < 1e-7) | (p_ > 1 - 1e-7))                        # every graph saturated
        if cce == "probs":
            assert not out["late"]["dp"].any() and not out["late"]["dz2"].any()   # clipped: no gradient at all


def test_head_inside_the_backward_refusals_and_fallbacks(ctx):
    """More than two classes: the backward launch refuses the in-kernel head (GCNX_ERR_UNSUPPORTED); gcnx_gemm_dw2(leaf)
    alone -- given the per-graph pool totals -- serves any class count through the head kernel's launch, and also when
    the weight gradients take their separate calls (too few rows for the split-K launch)."""
    from gcnx import device as D, synth
    from gcnx._lib import GcnxError
    from gcnx.device import DeviceCSR, Segments
    rng = np.random.default_rng(5)
    for n_graphs, c in ((3, 3), (1, 2)):
        hb = synth.ecoli_batch(n_graphs, 32, seed=c) if n_graphs > 1 else synth.power_law_batch(n_graphs=1, graph_size=96, f=32, seed=2, max_deg=24)
        n, b, h, f1 = hb.n, len(hb.graph_ptr) - 1, 32, 32
        a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr, symmetric=True)
        seg = Segments(ctx, hb.graph_ptr)
        y2 = ctx.to_device(np.maximum(rng.standard_normal((n, h), dtype=np.float32), 0))
        y1 = ctx.to_device(np.maximum(rng.standard_normal((n, f1), dtype=np.float32), 0))
        w2 = ctx.to_device((rng.standard_normal((f1, h)) / 6).astype(np.float32))
        w3 = ctx.to_device((rng.standard_normal((h, c)) / 6).astype(np.float32)); b3 = ctx.zeros(c)
        y = ctx.to_device(np.eye(c, dtype=np.float32)[rng.integers(0, c, b)])
        gpl = hb.graph_ptr
        psum = ctx.to_device(np.stack([y2.numpy()[gpl[g]:gpl[g + 1]].sum(0) for g in range(b)]).astype(np.float32))
        pcnt = ctx.to_device(np.stack([(y2.numpy()[gpl[g]:gpl[g + 1]] > 0).sum(0) for g in range(b)]).astype(np.float32))
        tiles = ctx.zeros((D.pool_tile_rows(n, b), h))
        ref = [ctx.zeros((b, h)), ctx.zeros((b, c)), ctx.zeros(2), ctx.zeros((h, c)), ctx.zeros(c), ctx.zeros((b, h)), ctx.zeros(h)]
        D.pool_dense_softmax_cce(ctx, seg, y2, ref[0], w3, b3, y, ref[1], ref[2], float(b), dw=ref[3], db=ref[4], dpooled=ref[5],
                                 mode="sum", db_relu=ref[6])
        got = [ctx.zeros(t.shape) for t in ref]
        ha = D.head_args(seg, tiles, tiles, psum, pcnt, w3, b3, y, float(b), got[1], got[2], got[3], got[4], got[6], got[0], got[5])
        dz2, dz1 = ctx.empty((n, h)), ctx.empty((n, f1))
        if c > 2:
            with pytest.raises(GcnxError, match="two classes"):
                D.gcn_conv_bwd_pool(ctx, a.transpose(), y2, seg, None, w2, y1, dz2, dz1, head=ha)
        D.gcn_conv_bwd_pool(ctx, a.transpose(), y2, seg, ref[5], w2, y1, dz2, dz1)
        s1 = ctx.to_device(rng.standard_normal((n, 32), dtype=np.float32))
        gw1, gw2 = ctx.empty((32, f1)), ctx.empty((32, h))
        D.gemm_dw2(ctx, s1, dz1, gw1, s1, dz2, gw2, leaf=ha)
        for r, g in zip(ref, got):
            assert rel_err(g.numpy(), r.numpy()) < TIGHT
        assert rel_err(gw2.numpy(), s1.numpy().astype(np.float64).T @ dz2.numpy()) < TIGHT


def test_gcn_conv_fused_edge_cases(ctx):
    """The one-launch GCNConv on degenerate inputs: fewer rows than a tile, rows without entries, single-node graphs, a
    graph boundary inside every tile, average pooling over graphs of one node -- forward, backward and the column sums."""
    from gcnx import device as D
    from gcnx.device import DeviceCSR, Segments
    o = O()
    rng = np.random.default_rng(3)
    sizes = np.array([1, 3, 1, 7, 2, 1, 40, 1], np.int64)             # 56 rows: two tiles, the second ragged
    gp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    n, b, f = int(gp[-1]), len(sizes), 32
    rows, cols = [], []
    for g in range(b):                                                # self-loops except on rows 4 and 20 (empty rows), a few in-graph pairs
        for i in range(gp[g], gp[g + 1]):
            if i in (4, 20):
                continue
            rows.append(i); cols.append(i)
            if sizes[g] > 2 and i + 1 < gp[g + 1]:
                rows += [i, i + 1]; cols += [i + 1, i]
    order = np.lexsort((cols, rows))
    rows, cols = np.asarray(rows)[order], np.asarray(cols)[order]
    keep = ~np.isin(rows, (4, 20))                                     # rows 4 and 20 end up with no entries at all
    rows, cols = rows[keep], cols[keep]
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    colidx = cols.astype(np.int32)
    vals = (rng.random(len(cols)) + 0.5).astype(np.float32)
    assert rowptr[5] == rowptr[4] and rowptr[21] == rowptr[20]
    a = DeviceCSR.from_host_csr(ctx, rowptr, colidx, vals, gp, symmetric=False)
    x = rng.standard_normal((n, f), dtype=np.float32)
    w = (rng.standard_normal((f, f)) / 6).astype(np.float32); bias = rng.standard_normal(f).astype(np.float32)
    out = ctx.empty((n, f)); s = ctx.empty((n, f))
    D.gcn_conv_fwd(ctx, a, ctx.to_device(x), ctx.to_device(w), ctx.to_device(bias), out, act="relu", s=s)
    ax = o.spmm_csr(rowptr.astype(np.int64), colidx.astype(np.int64), vals.astype(np.float64), x.astype(np.float64))
    assert rel_err(s.numpy(), ax) < TIGHT and not s.numpy()[[4, 20]].any()
    assert rel_err(out.numpy(), np.maximum(ax @ w.astype(np.float64) + bias, 0)) < TIGHT
    for mode in ("sum", "avg"):
        y2 = np.maximum(rng.standard_normal((n, f), dtype=np.float32), 0); y1 = np.maximum(rng.standard_normal((n, f), dtype=np.float32), 0)
        dp = rng.standard_normal((b, f), dtype=np.float32)
        rdz2 = o.global_pool_bwd(dp.astype(np.float64), gp, n, mode, None) * (y2 > 0)
        rdz1 = (o.spmm_csr_T(rowptr.astype(np.int64), colidx.astype(np.int64), vals.astype(np.float64), rdz2) @ w.astype(np.float64).T) * (y1 > 0)
        dz2 = ctx.empty((n, f)); dz1 = ctx.empty((n, f)); db1 = ctx.empty(f)
        D.gcn_conv_bwd_pool(ctx, a.transpose(), ctx.to_device(y2), Segments(ctx, gp), ctx.to_device(dp), ctx.to_device(w),
                            ctx.to_device(y1), dz2, dz1, db1=db1, mode=mode)
        assert rel_err(dz2.numpy(), rdz2) < TIGHT and rel_err(dz1.numpy(), rdz1) < TIGHT and rel_err(db1.numpy(), rdz1.sum(0)) < TIGHT
    # n smaller than one tile
    assert list(gp[:4]) == [0, 1, 4, 5]
    a5 = DeviceCSR.from_host_csr(ctx, rowptr[:6], colidx[:rowptr[5]], vals[:rowptr[5]], gp[:4].copy(), symmetric=False)
    out5 = ctx.empty((5, f))
    D.gcn_conv_fwd(ctx, a5, ctx.to_device(x[:5]), ctx.to_device(w), None, out5, act=None)
    assert rel_err(out5.numpy(), ax[:5] @ w.astype(np.float64)) < TIGHT


# ---- round 3 ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("bf16", None)])
@pytest.mark.parametrize("n,fi,fo", [(3000, 256, 256), (5000, 768, 256), (2100, 16, 256), (1000, 36, 48), (33, 128, 64)])
def test_gemm_wimage_panel_kernel_and_batch_norm_parts(ctx, n, fi, fo, prec, tol):
    """The split-bf16 panel GEMM of GeneralGNN's Dense layers (csrc/gemm_panel.hip; MatMul + BiasAdd under gcn.py:334): weight
    image of several matrices from ONE launch, out = x W + b against fp64 (bf16x3: 2e-5; plain bf16: exact against the
    bf16-rounded operands, loosely against fp64), ragged K (16, 36: zero-padded k steps), row counts that are not a
    multiple of the tile, and the batch-norm statistics of the output out of the epilogue -- (rows, mean, M2) parts merged
    by gcnx_bn_finalize_parts -- against numpy's two-pass moments and the Keras moving-statistics update."""
    from gcnx import device as D
    rng = np.random.default_rng(n + fi)
    x = (rng.standard_normal((n, fi)) * 2 + 0.5).astype(np.float32)
    w = (rng.standard_normal((fi, fo)) / np.sqrt(fi)).astype(np.float32)
    w2 = (rng.standard_normal((fi, fo)) / np.sqrt(fi)).astype(np.float32)
    b = rng.standard_normal(fo).astype(np.float32)
    dx, dw, dw2, db = ctx.to_device(x), ctx.to_device(w), ctx.to_device(w2), ctx.to_device(b)
    img = ctx.empty(D.wimage_elems(ctx, fi, fo, False, prec), np.uint16)
    img2 = ctx.empty(D.wimage_elems(ctx, fi, fo, False, prec), np.uint16)
    D.wimage_prepare(ctx, [(dw, img, False, prec), (dw2, img2, False, prec)])
    out = ctx.zeros((n, fo))
    parts = ctx.zeros(3 * max(D.gemm_wimage_parts(ctx, n), 1) * fo)
    nparts = D.gemm_wimage(ctx, dx, img, fi, fo, out, bias=db, prec=prec, bn_parts=parts)
    assert nparts is not None and nparts >= 1
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    got = out.numpy()
    if tol is not None:
        assert rel_err(got, ref) < tol
    else:
        o = O()
        rb = lambda v: o.bf16_from_bits(o.bf16_bits(v)).astype(np.float64)
        assert rel_err(got, rb(x) @ rb(w) + b) < 1e-5 and rel_err(got, ref) < 2e-2
    # the second matrix of the same prepare launch
    out2 = ctx.zeros((n, fo))
    assert D.gemm_wimage(ctx, dx, img2, fi, fo, out2, prec=prec) == 0
    assert rel_err(out2.numpy(), x.astype(np.float64) @ w2.astype(np.float64)) < (tol or 2e-2)
    # batch-norm moments of what was written
    mean, inv = ctx.empty(fo), ctx.empty(fo)
    mm, mv = ctx.to_device(np.full(fo, 0.25, np.float32)), ctx.to_device(np.full(fo, 2.0, np.float32))
    D.bn_finalize_parts(ctx, parts, nparts, mean, inv, mm, mv)
    g64 = got.astype(np.float64)
    assert rel_err(mean.numpy(), g64.mean(0)) < TIGHT
    assert rel_err(inv.numpy(), 1.0 / np.sqrt(g64.var(0) + 1e-3)) < TIGHT
    assert rel_err(mm.numpy(), 0.99 * 0.25 + 0.01 * g64.mean(0)) < TIGHT and rel_err(mv.numpy(), 0.99 * 2.0 + 0.01 * g64.var(0)) < TIGHT
    # shapes the kernel does not serve are refused without a launch (and without a message)
    xm = ctx.zeros((n, fi + 4)).cols(1, fi + 1)                       # a 4-byte-aligned operand
    assert D.gemm_wimage(ctx, xm, img, fi, fo, out, prec=prec) is None


@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("n,fi", [(4097, 256), (5000, 512), (3001, 1024)])
def test_gemm_wimage_transposed_accumulate_and_dw_panels(ctx, n, fi, prec, tol):
    """The two backward products of a GeneralGNN Dense layer on the panel kernels: dX (+)= dH W^T through the transposed image
    (A-resident wide kernel for fi > 256: the column panels walked inside the workgroup; accumulate = the skip
    connections' gradient sum), and dW = X^T dH as 256-column panels of the streaming kernel with one reduction launch --
    on column-slice views of wider buffers (the concat buffer of GeneralGNN.call), against fp64."""
    from gcnx import device as D
    rng = np.random.default_rng(fi + n)
    fo = 256
    w = (rng.standard_normal((fi, fo)) / 16).astype(np.float32)
    dh = rng.standard_normal((n, fo)).astype(np.float32)
    cat = rng.standard_normal((n, fi + 64)).astype(np.float32)            # x and dx are column slices [32, 32 + fi)
    dcat0 = rng.standard_normal((n, fi + 64)).astype(np.float32)
    dw_, ddh, dcat, ddcat = ctx.to_device(w), ctx.to_device(dh), ctx.to_device(cat), ctx.to_device(dcat0)
    imgt = ctx.empty(D.wimage_elems(ctx, fi, fo, True, prec), np.uint16)
    D.wimage_prepare(ctx, [(dw_, imgt, True, prec)])
    xs, dxs = dcat.cols(32, 32 + fi), ddcat.cols(32, 32 + fi)
    assert D.gemm_wimage(ctx, ddh, imgt, fi, fo, dxs, transpose=True, prec=prec, accumulate=True) == 0
    want = dcat0.astype(np.float64)
    want[:, 32:32 + fi] += dh.astype(np.float64) @ w.astype(np.float64).T
    got = ddcat.numpy()
    assert rel_err(got, want) < tol
    assert np.array_equal(got[:, :32], dcat0[:, :32]) and np.array_equal(got[:, 32 + fi:], dcat0[:, 32 + fi:])   # neighbours untouched
    assert D.gemm_wimage(ctx, ddh, imgt, fi, fo, dxs, transpose=True, prec=prec) == 0                              # overwrite form
    assert rel_err(ddcat.numpy()[:, 32:32 + fi], dh.astype(np.float64) @ w.astype(np.float64).T) < tol
    g = ctx.empty((fi, fo))
    D.gemm_dw(ctx, xs, ddh, g, prec=prec)
    assert rel_err(g.numpy(), cat[:, 32:32 + fi].astype(np.float64).T @ dh.astype(np.float64)) < tol
    g2 = ctx.empty((fi, fo)); D.gemm_dw(ctx, xs, ddh, g2, prec=prec)
    assert np.array_equal(g.numpy(), g2.numpy())                                                                    # deterministic


@pytest.mark.parametrize("weighted", [False, True])
def test_spmm_hub_rows_as_segments_on_their_own_workgroups(ctx, weighted):
    """BASELINE config 5's shape in small: 16 power-law graphs of 8 192 nodes (degrees up to 4096), F = 64.  A batch of this
    size gets a plan although it has few graphs; the plan lists the rows of more than 256 entries as 256-entry segments
    (spmm_hub_seg_kernel -> spmm_hub_combine_kernel), the row gather skips them -- forward aggregation with bias + ReLU and
    the folded pool backward against scipy in fp64, bit-reproducible."""
    import scipy.sparse as sp
    from gcnx import device as D, synth
    from gcnx.device import Segments
    hb = synth.power_law_batch(16, 8192, 64, seed=3)
    deg = np.diff(hb.rowptr)
    assert hb.n >= 131072 and deg.max() >= 2048 and (deg > 256).sum() >= 16
    csr, vals = _csr(ctx, hb, weighted)
    assert csr.plan is not None
    rng = np.random.default_rng(5)
    f = 64
    a64 = sp.csr_matrix((np.ones(hb.nnz) if vals is None else vals.astype(np.float64), hb.colidx, hb.rowptr), shape=(hb.n, hb.n))
    h = rng.standard_normal((hb.n, f), dtype=np.float32); bias = rng.standard_normal(f).astype(np.float32)
    out = ctx.zeros((hb.n, f))
    D.spmm(ctx, csr, ctx.to_device(h), ctx.to_device(bias), out, act="relu")
    ref = np.maximum(a64 @ h.astype(np.float64) + bias, 0)
    got = out.numpy()
    assert rel_err(got, ref) < TIGHT
    hubs = np.nonzero(deg > 256)[0]
    assert rel_err(got[hubs], ref[hubs]) < TIGHT                      # the hub rows themselves
    out2 = ctx.zeros((hb.n, f)); D.spmm(ctx, csr, ctx.to_device(h), ctx.to_device(bias), out2, act="relu")
    assert np.array_equal(got, out2.numpy())
    # folded pool backward (A symmetric here: the same operator)
    y = np.maximum(rng.standard_normal((hb.n, f), dtype=np.float32), 0)
    dp = rng.standard_normal((hb.n_graphs, f), dtype=np.float32)
    seg = Segments(ctx, hb.graph_ptr)
    o2 = ctx.zeros((hb.n, f))
    D.spmm_pool_bwd(ctx, csr, ctx.to_device(y), seg, ctx.to_device(dp), o2, "avg")
    sizes = np.diff(hb.graph_ptr)
    dz = np.repeat(dp.astype(np.float64) / sizes[:, None], sizes, axis=0) * (y > 0)
    assert rel_err(o2.numpy(), a64.T @ dz) < TIGHT


def test_bf16_storage_between_bf16_operand_gemms_changes_no_bit(ctx):
    """bf16 STORAGE (r3, include/gcnx.h): gcnx_gemm_fwd_bf16 / gcnx_gemm_dx_bf16 / gcnx_gemm_dw_bf16 read their streamed
    operands as bfloat16 and (the first two) store the result rounded to bfloat16.  Against the fp32-storage entry points
    with GCNX_PREC_BF16 on the SAME values: every product, bias gradient and bit image identical bit for bit, stored results
    equal to the RNE rounding (oracle.bf16_bits) of the fp32 ones.  Ragged row count, zero rows past the end."""
    from gcnx import device as D
    o = O()
    rng = np.random.default_rng(21)
    n = 40_003
    x32 = o.bf16_from_bits(o.bf16_bits(rng.standard_normal((n, 256), dtype=np.float32)))      # values bf16 holds exactly
    w = (rng.standard_normal((256, 256)) / 16).astype(np.float32)
    b = rng.standard_normal(256).astype(np.float32)
    dx, dw_, dbias = ctx.to_device(x32), ctx.to_device(w), ctx.to_device(b)
    x16 = D.to_bf16(ctx, dx)
    assert np.array_equal(x16.numpy(), o.bf16_bits(x32))
    # forward with ReLU + bit image: fp32 storage (gcnx_gemm_relu_bits) against bf16 in / bf16 out
    y32 = ctx.empty((n, 256)); bits32 = ctx.zeros(n * 16, np.int32)
    assert D.gemm_relu_bits(ctx, dx, dw_, dbias, y32, bits32, prec="bf16")
    y16 = ctx.empty((n, 256), np.uint16); bits16 = ctx.zeros(n * 16, np.int32)
    assert D.gemm_fwd_bf16(ctx, x16, dw_, dbias, y16, act="relu", bits=bits16)
    assert np.array_equal(y16.numpy(), o.bf16_bits(y32.numpy()))
    assert np.array_equal(bits16.numpy(), bits32.numpy())
    # ... against the oracle's bf16-operand product (fp64 accumulation of the rounded operands)
    ref = np.maximum(x32.astype(np.float64) @ o.bf16_from_bits(o.bf16_bits(w)).astype(np.float64) + b, 0)
    assert rel_err(o.bf16_from_bits(y16.numpy()), ref) < 2.0 ** -8
    # forward without activation, fp32 result (layer 2's kernel product: the aggregation reads it in fp32)
    h32 = ctx.empty((n, 256)); h32b = ctx.empty((n, 256))
    D.gemm(ctx, dx, dw_, None, h32, prec="bf16")
    assert D.gemm_fwd_bf16(ctx, x16, dw_, None, h32b)
    assert np.array_equal(h32.numpy(), h32b.numpy())
    # dX with the bit image and the bias gradient of the layer below
    dz32 = ctx.empty((n, 256)); db32 = ctx.zeros(256)
    D.gemm_dx(ctx, dx, dw_, dz32, prec="bf16", y_mask=y32, db=db32, mask_bits=bits32)
    dz16 = ctx.empty((n, 256), np.uint16); db16 = ctx.zeros(256)
    assert D.gemm_dx_bf16(ctx, x16, dw_, dz16, mask_bits=bits16, db=db16)
    assert np.array_equal(dz16.numpy(), o.bf16_bits(dz32.numpy()))
    assert np.array_equal(db16.numpy(), db32.numpy())
    # dW from two bf16-stored operands
    g32 = ctx.empty((256, 256)); g16 = ctx.empty((256, 256))
    dzr = ctx.to_device(o.bf16_from_bits(dz16.numpy()))
    D.gemm_dw(ctx, dx, dzr, g32, prec="bf16")
    assert D.gemm_dw_bf16(ctx, x16, dz16, g16)
    assert np.array_equal(g16.numpy(), g32.numpy())
    # the weight images of a whole step from ONE launch (gcnx_gemm_stream_images) instead of one small launch per product
    img_store = ctx.empty(2 * 65536, np.uint16)                # (the caller owns the images: they must outlive the products)
    imgs = D.stream_images(ctx, [(dw_, True), (dw_, False)], img_store)
    y16b = ctx.zeros((n, 256), np.uint16); dz16b = ctx.zeros((n, 256), np.uint16); db16b = ctx.zeros(256)
    assert D.gemm_fwd_bf16(ctx, x16, dw_, dbias, y16b, act="relu", wimg=imgs[0])
    assert D.gemm_dx_bf16(ctx, x16, dw_, dz16b, mask_bits=bits16, db=db16b, wimg=imgs[1])
    assert np.array_equal(y16b.numpy(), y16.numpy()) and np.array_equal(dz16b.numpy(), dz16.numpy()) and np.array_equal(db16b.numpy(), db16.numpy())
    # shapes outside the streaming kernels: an answer, nothing launched
    small = ctx.zeros((1000, 256), np.uint16)
    assert not D.gemm_fwd_bf16(ctx, small, dw_, None, ctx.empty((1000, 256)))
    assert not D.gemm_dw_bf16(ctx, small, small, g16)


@pytest.mark.parametrize("mode", ["sum", "avg"])
def test_spmm_bf16_result_rows_on_the_tile_kernels(ctx, mode):
    """gcnx_spmm_csr_bf16out / gcnx_spmm_csr_pool_bwd_bf16out: the tile kernel (double-buffered and single tiles) and the
    8-row chunks of taller graphs store the rounded row -- equal to oracle.bf16_bits of the fp32 entry point's result, which
    the tile tests hold to the oracle -- and the forms refuse what they do not serve."""
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR, Segments
    o = O()
    hb = synth.block_diag_batch(150_000, 1_500_000, 256, seed=6, with_x=False)
    sizes = np.diff(hb.graph_ptr)
    assert len(sizes) >= 128 and (sizes <= 624).any() and ((sizes > 624) & (sizes <= 1276)).any() and (sizes > 1276).any()
    n, f, b = hb.n, 256, len(sizes)
    rng = np.random.default_rng(9)
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
    assert a.plan is not None
    x = ctx.to_device(rng.standard_normal((n, f), dtype=np.float32)); bias = ctx.to_device(rng.standard_normal(f).astype(np.float32))
    out32 = ctx.empty((n, f)); out16 = ctx.zeros((n, f), np.uint16)
    for act, bb in ((None, None), ("relu", bias)):
        D.spmm(ctx, a, x, bb, out32, act=act)
        assert D.spmm_bf16out(ctx, a, x, bb, out16, act=act)
        assert np.array_equal(out16.numpy(), o.bf16_bits(out32.numpy())), act
    # ... independently: within one bf16 rounding of the fp64 aggregation
    ref = _ref_spmm(hb, vals, x.numpy(), None, False)
    D.spmm_bf16out(ctx, a, x, None, out16)
    assert rel_err(o.bf16_from_bits(out16.numpy()), ref) < 2.0 ** -8
    # the folded backward aggregation from the bit image
    seg = Segments(ctx, hb.graph_ptr)
    y2 = ctx.empty((n, f)); bits = ctx.zeros((f // 32) * n, np.int32)
    assert D.spmm_relu_bits(ctx, a, x, bias, y2, bits)
    dp = ctx.to_device(rng.standard_normal((b, f), dtype=np.float32))
    at = a.transpose()
    D.spmm_pool_bwd(ctx, at, y2, seg, dp, out32, mode, y_bits=bits)
    assert D.spmm_pool_bwd_bf16out(ctx, at, y2, seg, dp, out16, mode, y_bits=bits)
    assert np.array_equal(out16.numpy(), o.bf16_bits(out32.numpy()))
    # refusals: no bit image; an unweighted operator; the row gather forced
    assert not D.spmm_pool_bwd_bf16out(ctx, at, y2, seg, dp, out16, mode, y_bits=None)
    au = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    assert not D.spmm_bf16out(ctx, au, x, None, out16)
    try:
        ctx.set_tuning("spmm_kernel", "rows")
        assert not D.spmm_bf16out(ctx, a, x, None, out16)
    finally:
        ctx.set_tuning("spmm_kernel", "auto")


@pytest.mark.parametrize("mode", ["sum", "avg"])
def test_spmm_relu_bits_pool_leaves_bits_pooled_rows_and_counts_without_the_output(ctx, mode):
    """gcnx_spmm_csr_relu_bits_pool (r3): against the two-launch form -- gcnx_spmm_csr_relu_bits, then the global pool and the
    positive counts of its output.  The bit image is identical; pooled rows agree to summation order (the tile kernel adds a
    graph's rows per lane, per wave, then over its 16 waves), counts exactly; rows of `out` are written for the graphs taller
    than a tile only (identical there, untouched elsewhere); double-buffered and single tiles, tall graphs in the batch."""
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR, Segments
    o = O()
    hb = synth.block_diag_batch(150_000, 1_500_000, 256, seed=6, with_x=False)
    sizes = np.diff(hb.graph_ptr)
    assert len(sizes) >= 128 and (sizes <= 624).any() and ((sizes > 624) & (sizes <= 1276)).any() and (sizes > 1276).any()
    n, f, b = hb.n, 256, len(sizes)
    rng = np.random.default_rng(10)
    vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
    seg = Segments(ctx, hb.graph_ptr)
    h = ctx.to_device(rng.standard_normal((n, f), dtype=np.float32)); bias = ctx.to_device(rng.standard_normal(f).astype(np.float32))
    y_ref = ctx.empty((n, f)); bits_ref = ctx.zeros((f // 32) * n, np.int32)
    assert D.spmm_relu_bits(ctx, a, h, bias, y_ref, bits_ref)
    yh = y_ref.numpy()
    pooled_ref, _ = o.global_pool_fwd(yh.astype(np.float64), hb.graph_ptr, mode)
    cnt_ref = np.add.reduceat((yh > 0).astype(np.float64), hb.graph_ptr[:-1].astype(np.int64), axis=0)
    sentinel = np.float32(-777.0)
    y = ctx.to_device(np.full((n, f), sentinel, np.float32)); bits = ctx.zeros((f // 32) * n, np.int32)
    pooled = ctx.zeros((b, f)); cnt = ctx.zeros((b, f))
    assert D.spmm_relu_bits_pool(ctx, a, h, bias, y, bits, seg, pooled, cnt, mode)
    tile_rows = np.concatenate([np.arange(hb.graph_ptr[g], hb.graph_ptr[g + 1]) for g in range(b) if sizes[g] <= 1276])
    tall_rows = np.concatenate([np.arange(hb.graph_ptr[g], hb.graph_ptr[g + 1]) for g in range(b) if sizes[g] > 1276])
    img, img_ref = bits.numpy().reshape(f // 32, n), bits_ref.numpy().reshape(f // 32, n)
    assert np.array_equal(img[:, tile_rows], img_ref[:, tile_rows])
    got_y = y.numpy()
    assert np.all(got_y[tile_rows] == sentinel) and np.array_equal(got_y[tall_rows], yh[tall_rows])
    assert np.array_equal(cnt.numpy(), cnt_ref)
    assert rel_err(pooled.numpy(), pooled_ref) < TIGHT
    p2 = ctx.zeros((b, f)); c2 = ctx.zeros((b, f))                      # deterministic
    assert D.spmm_relu_bits_pool(ctx, a, h, bias, y, bits, seg, p2, c2, mode)
    assert np.array_equal(p2.numpy(), pooled.numpy())
    try:                                                                # refused when the tile kernels are switched off
        ctx.set_tuning("spmm_kernel", "rows")
        assert not D.spmm_relu_bits_pool(ctx, a, h, bias, y, bits, seg, p2, c2, mode)
    finally:
        ctx.set_tuning("spmm_kernel", "auto")


def test_round3_entry_points_refuse_bad_arguments_loudly(ctx):
    """Error behaviour of the entry points added in round 3: a wrong argument is an error with a message (GcnxError), a shape
    the kernels do not serve is an ANSWER (False from the wrapper, nothing launched, gcnx_last_error untouched)."""
    import ctypes as C
    from gcnx import device as D, _lib as L, synth
    from gcnx._lib import GcnxError
    from gcnx.device import DeviceCSR, Segments
    w = ctx.zeros((256, 256)); store = ctx.empty(65536, np.uint16)
    # gcnx_gemm_stream_images: only 256 x 256 operands, at most four jobs, aligned images
    jobs = (L.StreamImageJob * 1)(L.StreamImageJob(ctx.zeros((128, 256)).ptr, 128, 256, 1, store.ptr))
    with pytest.raises(GcnxError, match="256 x 256"):
        ctx._ck(ctx.lib.gcnx_gemm_stream_images(ctx.h, 1, C.cast(jobs, C.c_void_p)))
    jobs = (L.StreamImageJob * 1)(L.StreamImageJob(w.ptr, 256, 256, 1, store.ptr + 2))
    with pytest.raises(GcnxError, match="aligned"):
        ctx._ck(ctx.lib.gcnx_gemm_stream_images(ctx.h, 1, C.cast(jobs, C.c_void_p)))
    with pytest.raises(GcnxError, match="0 .. 4"):
        ctx._ck(ctx.lib.gcnx_gemm_stream_images(ctx.h, 5, C.cast(jobs, C.c_void_p)))
    # the pooled head: db_relu needs the counts
    hb = synth.ecoli_batch(4, 32, seed=2)
    seg = Segments(ctx, hb.graph_ptr)
    pooled = ctx.zeros((4, 32)); w3 = ctx.zeros((32, 2)); yv = ctx.to_device(hb.y.astype(np.float32))
    probs = ctx.empty((4, 2)); la = ctx.zeros(2); dw = ctx.empty((32, 2)); db = ctx.empty(2); dp = ctx.empty((4, 32)); dbr = ctx.empty(32)
    with pytest.raises(GcnxError, match="counts"):
        D.pooled_dense_softmax_cce(ctx, seg, pooled, None, w3, None, yv, probs, la, 4.0, dw=dw, db=db, dpooled=dp, db_relu=dbr)
    D.pooled_dense_softmax_cce(ctx, seg, pooled, None, w3, None, yv, probs, la, 4.0)      # forward only: fine
    assert np.allclose(probs.numpy(), 0.5)
    # the pooled layer without its output: a batch without a tile plan is an answer, not an error
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, synth.gcn_norm_host(hb.rowptr, hb.colidx), hb.graph_ptr)
    n = hb.n
    h = ctx.zeros((n, 32)); y = ctx.empty((n, 32)); bits = ctx.zeros(n, np.int32); cnt = ctx.zeros((4, 32))
    err_before = ctx.lib.gcnx_last_error(ctx.h)
    assert not D.spmm_relu_bits_pool(ctx, a, h, None, y, bits, seg, pooled, cnt, "sum")
    assert ctx.lib.gcnx_last_error(ctx.h) == err_before
    assert not D.spmm_bf16out(ctx, a, h, None, ctx.empty((n, 32), np.uint16))
    assert not D.gemm_dx_bf16(ctx, ctx.zeros((1000, 256), np.uint16), w, ctx.empty((1000, 256), np.uint16))


@pytest.mark.parametrize("weighted", [False, True])
def test_spmm_double_buffered_tiles_with_rows_past_the_register_held_entries(ctx, weighted):
    """in a double-buffered unit (graphs of <= 624 rows on the 1024-thread tile shape) the on-demand
    fetch of a row's entries past the 32 / 16 held in registers must pad with the CURRENT buffer's zero row -- it padded with
    the single-tile zero row, an LDS address outside the allocation when the unit sits in the second buffer.  Many graphs of
    100..600 rows, each with a few rows of 40..250 entries, f = 256 (eight slabs per unit: every unit prefetches): the tile
    kernels against the row gather, which adds a row's entries in the same (CSR) order -- bit for bit -- and the oracle."""
    from gcnx import device as D, synth
    import scipy.sparse as sp
    rng = np.random.default_rng(21)
    sizes = rng.integers(100, 601, size=160)
    sizes[:4] = (624, 600, 101, 320)
    blocks = []
    for i, m in enumerate(sizes):
        a = sp.random(m, m, density=min(1.0, 8.0 / m), random_state=i, format="lil")
        for r in rng.choice(m, size=3, replace=False):                 # a few long rows per graph
            deg = int(rng.integers(40, min(230, m - 1) + 1))
            a[r, rng.choice(m, size=deg, replace=False)] = 1.0
        a = a.tocsr()
        a = ((a + a.T) > 0).astype(np.float32) + sp.identity(m, dtype=np.float32, format="csr")
        blocks.append((a > 0).astype(np.float32))
    a = sp.block_diag(blocks).tocsr(); a.sort_indices()
    n = a.shape[0]
    gp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    f = 256
    hb = synth.HostBatch(rng.standard_normal((n, f), dtype=np.float32), a.indptr.astype(np.int32), a.indices.astype(np.int32),
                         None, gp, np.zeros((len(sizes), 2), np.float32))
    deg = np.diff(hb.rowptr)
    # (<= 256 entries: longer rows the row gather splits over its four waves -- another summation order)
    assert 200 < deg.max() <= 256 and (deg > 32).sum() >= 3 * len(sizes)
    csr, vals = _csr(ctx, hb, weighted)
    assert csr.plan is not None
    bias = rng.standard_normal(f).astype(np.float32)
    h, db = ctx.to_device(hb.x), ctx.to_device(bias)
    outs = {}
    try:
        for kernel in ("tile", "rows"):
            ctx.set_tuning("spmm_kernel", kernel)
            o = ctx.zeros((n, f))
            D.spmm(ctx, csr, h, db, o, act="relu")
            outs[kernel] = o.numpy()
        # the bit-image form (kDuoBitsOut: double-buffered too) writes the same rows
        ctx.set_tuning("spmm_kernel", "tile")
        o = ctx.zeros((n, f))
        bits = ctx.zeros((f // 32) * n, np.int32)
        assert D.spmm_relu_bits(ctx, csr, h, db, o, bits)
        outs["bits"] = o.numpy()
    finally:
        ctx.set_tuning("spmm_kernel", "auto")
    assert np.array_equal(outs["tile"], outs["rows"])
    assert np.array_equal(outs["bits"], outs["rows"])
    assert rel_err(outs["tile"], _ref_spmm(hb, vals, hb.x, bias, True)) < TIGHT


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("f", [128, 192, 256, 512])
def test_spmm_column_blocks_for_graphs_larger_than_an_xcd_l2(ctx, f, weighted):
    """BASELINE config 5 in small (r4): power-law graphs of 8 192 nodes (degrees up to 4096) are walked by spmm_cb_kernel one
    64-column block at a time -- rows of at most 32 entries sixteen to a wave (four per lane group), longer rows a wave each,
    hub rows by all four waves -- next to graphs of ordinary size, which keep the row gather (this batch) or the tile kernels (the next test).
    Against scipy in fp64; at f = 256 the rows of at most 32 entries bit for bit against the row gather + hub segments of
    round 3 for the rows of at most 32 entries (GCNX_SPMM_CB=0: same CSR-order sums); bit-reproducible; bias + ReLU and plain."""
    import scipy.sparse as sp
    from gcnx import device as D, synth
    big = synth.power_law_batch(3, 8192, f, seed=3)
    small = synth.ecoli_batch(5, f, seed=8)
    hb = synth.concat_batches([small.slice_graphs(0, 2), big, small.slice_graphs(2, 5)])     # cb graphs in the middle
    deg = np.diff(hb.rowptr)
    assert deg.max() >= 2048 and (deg > 256).sum() >= 3 and (np.diff(hb.graph_ptr) >= 4096).sum() == 3
    csr, vals = _csr(ctx, hb, weighted)
    assert csr.plan is not None
    rng = np.random.default_rng(7)
    a64 = sp.csr_matrix((np.ones(hb.nnz) if vals is None else vals.astype(np.float64), hb.colidx, hb.rowptr), shape=(hb.n, hb.n))
    h = rng.standard_normal((hb.n, f), dtype=np.float32); bias = rng.standard_normal(f).astype(np.float32)
    dh, db = ctx.to_device(h), ctx.to_device(bias)
    outs = {}
    try:
        for cb in (1, 0):
            ctx.set_tuning("spmm_cb", cb)
            o = ctx.zeros((hb.n, f))
            D.spmm(ctx, csr, dh, db, o, act="relu")
            outs[cb] = o.numpy()
        ctx.set_tuning("spmm_cb", 1)
        o2 = ctx.zeros((hb.n, f)); D.spmm(ctx, csr, dh, db, o2, act="relu")
        assert np.array_equal(o2.numpy(), outs[1])                                  # bit-reproducible
        o3 = ctx.zeros((hb.n, f)); D.spmm(ctx, csr, dh, None, o3)
        assert rel_err(o3.numpy(), a64 @ h.astype(np.float64)) < TIGHT              # no bias, no activation
    finally:
        ctx.set_tuning("spmm_cb", 1)
    ref = np.maximum(a64 @ h.astype(np.float64) + bias, 0)
    assert rel_err(outs[1], ref) < TIGHT and rel_err(outs[0], ref) < TIGHT
    hubs = np.nonzero(deg > 256)[0]
    assert rel_err(outs[1][hubs], ref[hubs]) < TIGHT
    if f == 256:       # (at f = 256 the row gather walks a row with ONE lane group, in CSR order like the column blocks; at 128 with two)
        short = deg <= 32                        # (kCbShort: the rows a lane group walks alone)
        assert np.array_equal(outs[1][short], outs[0][short])


@pytest.mark.parametrize("mode", ["sum", "avg"])
@pytest.mark.parametrize("weighted", [False, True])
def test_spmm_pool_bwd_fold_in_column_blocks(ctx, mode, weighted):
    """The folded pool' / ReLU' backward aggregation (gcnx_spmm_csr_pool_bwd) on graphs of >= 4096 rows (r4): the column-block
    kernel gathers the saved ReLU output as its 0 / 1 mask and multiplies the finished sums by the graph's dPooled row (x 1 / n_g
    for the average pool) -- against A^T (pool'(dPooled) * [y > 0]) in fp64, against the r3 path (row gather + hub segments,
    GCNX_SPMM_CB=0) on the same operands, bit-reproducible; power-law graphs (hub rows included) between graphs of ordinary size."""
    import scipy.sparse as sp
    from gcnx import device as D, synth
    from gcnx.device import Segments
    f = 256
    big = synth.power_law_batch(2, 8192, f, seed=4)
    small = synth.ecoli_batch(4, f, seed=9)
    hb = synth.concat_batches([small.slice_graphs(0, 1), big, small.slice_graphs(1, 4)])
    csr, vals = _csr(ctx, hb, weighted)
    assert csr.plan is not None and (np.diff(hb.graph_ptr) >= 4096).sum() == 2
    rng = np.random.default_rng(3)
    y = np.maximum(rng.standard_normal((hb.n, f), dtype=np.float32), 0)           # a ReLU output: half zeros
    dp = rng.standard_normal((hb.n_graphs, f)).astype(np.float32)
    a64 = sp.csr_matrix((np.ones(hb.nnz) if vals is None else vals.astype(np.float64), hb.colidx, hb.rowptr), shape=(hb.n, hb.n))
    ng = np.diff(hb.graph_ptr)
    scale = (1.0 / ng if mode == "avg" else np.ones(hb.n_graphs))
    drow = np.repeat(dp.astype(np.float64) * scale[:, None], ng, axis=0)
    ref = (a64.T @ (y > 0).astype(np.float64)) * drow                             # (the operator is symmetric: A^T = A)
    seg, dy, ddp = Segments(ctx, hb.graph_ptr), ctx.to_device(y), ctx.to_device(dp)
    outs = {}
    try:
        for cb in (1, 0):
            ctx.set_tuning("spmm_cb", cb)
            o = ctx.zeros((hb.n, f))
            D.spmm_pool_bwd(ctx, csr.transpose(), dy, seg, ddp, o, mode)
            outs[cb] = o.numpy()
        ctx.set_tuning("spmm_cb", 1)
        o2 = ctx.zeros((hb.n, f)); D.spmm_pool_bwd(ctx, csr.transpose(), dy, seg, ddp, o2, mode)
        assert np.array_equal(o2.numpy(), outs[1])
    finally:
        ctx.set_tuning("spmm_cb", 1)
    assert rel_err(outs[1], ref) < TIGHT and rel_err(outs[0], ref) < TIGHT
    short = np.diff(hb.rowptr) <= 32
    assert rel_err(outs[1][short], outs[0][short]) < 1e-6        # same CSR-order sums; the dPooled scale is applied in one rounding either way


def test_spmm_column_blocks_next_to_the_tile_kernels(ctx):
    """A batch with enough small graphs for the tile kernels AND two graphs of >= 4096 rows: tiles for the former, column
    blocks for the latter, one call -- against scipy, and the captured form replays (the work list is built by the eager run)."""
    import scipy.sparse as sp
    from gcnx import device as D, synth
    f = 256
    big = synth.power_law_batch(2, 4608, f, seed=5, max_deg=1024)
    small = synth.block_diag_batch(60000, 600000, f, seed=4, mean_size=300)
    hb = synth.concat_batches([small, big])
    assert small.n_graphs >= 128 and (np.diff(hb.graph_ptr) >= 4096).sum() == 2
    csr, vals = _csr(ctx, hb, True)
    a64 = sp.csr_matrix((vals.astype(np.float64), hb.colidx, hb.rowptr), shape=(hb.n, hb.n))
    rng = np.random.default_rng(9)
    h = rng.standard_normal((hb.n, f), dtype=np.float32); bias = rng.standard_normal(f).astype(np.float32)
    dh, db = ctx.to_device(h), ctx.to_device(bias)
    o = ctx.zeros((hb.n, f))
    ctx.set_tuning("spmm_cb", 1)                       # (the default since the index loads of the kernel became few and wide)
    try:
        D.spmm(ctx, csr, dh, db, o, act="relu")
        ref = np.maximum(a64 @ h.astype(np.float64) + bias, 0)
        assert rel_err(o.numpy(), ref) < TIGHT
        o.fill_zero()
        g = ctx.capture(lambda: D.spmm(ctx, csr, dh, db, o, act="relu"))
        g.launch(); ctx.sync()
        assert rel_err(o.numpy(), ref) < TIGHT
        g.destroy()
    finally:
        ctx.set_tuning("spmm_cb", 1)
