"""Models on the hot path.

``GCN2``: GCNConv(F->H, relu) -> GCNConv(H->H, relu) -> GlobalSumPool -> Dense(H->C) softmax,
categorical cross-entropy, SGD -- the "2-layer GCNConv + global pool" topology the reference
defines at gcn_utills.py:805-808,832-842 written with the Spektral layers the live script uses,
driven exactly like the reference's train_step / evaluate (src/scripts/gcn.py:328-340,
342-362): forward, CategoricalCrossentropy, gradients of every trainable variable, SGD apply,
mean categorical accuracy.  BN-free, so an N-GPU step equals the 1-GPU step.

The step is a fixed sequence of libgcnx calls on preallocated buffers; after one eager run it
is captured into a HIP graph (the role tf.function plays at gcn.py:328) and replayed.
"""
from __future__ import annotations

import numpy as np

from . import device as D
from .layers import glorot_uniform
from .loader import SparseTensor


class DeviceBatch:
    """A DisjointLoader batch resident in HBM: x [N,F] fp32, adjacency CSR, graph segments,
    one-hot labels y [B,C] fp32."""

    _next_uid = 0

    def __init__(self, ctx, x, a, seg, y=None):
        self.ctx, self.x, self.a, self.seg, self.y = ctx, x, a, seg, y
        DeviceBatch._next_uid += 1
        self.uid = DeviceBatch._next_uid  # never reused (unlike id()): keys captured graphs
        self.n, self.f = x.shape
        self.n_graphs = seg.n_graphs

    @classmethod
    def from_host(cls, ctx, inputs, y=None, normalize=None, weighted=True, symmetric=True):
        """inputs = (x, a, i) as yielded by DisjointLoader.  ``a`` is a SparseTensor (COO) or a
        scipy sparse matrix.  normalize='spektral'|'pyg' applies gcn_filter on the device (the
        CSR must then hold every diagonal entry, as the reference's self-looped graphs do)."""
        x, a, i = inputs
        n = x.shape[0]
        seg = i if isinstance(i, D.Segments) else D.Segments.from_ids(ctx, i)
        dx = ctx.to_device(x, np.float32)
        if isinstance(a, D.DeviceCSR):
            csr = a
        else:
            if not isinstance(a, SparseTensor):
                from .loader import sp_matrix_to_sp_tensor
                a = sp_matrix_to_sp_tensor(a)
            csr = D.DeviceCSR.from_coo(ctx, a.indices, a.values, n, graph_ptr=seg.host, symmetric=symmetric,
                                       weighted=weighted)
        if normalize:
            csr = csr.gcn_norm(normalize)
        dy = ctx.to_device(y, np.float32) if y is not None else None
        return cls(ctx, dx, csr, seg, dy)


class GCN2:
    PARAM_ORDER = ("w1", "b1", "w2", "b2", "w3", "b3")

    def __init__(self, ctx, n_labels=2, hidden=None, pool="sum", prec="f32", seed=0, comm=None, use_graph=True):
        self.ctx, self.n_labels, self.hidden, self.pool, self.prec = ctx, int(n_labels), hidden, pool, prec
        self.comm = comm                     # gcnx.comm.Communicator or None
        self.use_graph = use_graph
        self._rng = np.random.default_rng(seed)
        self.built = False
        self._bufs = None
        self._graphs = {}

    # ---- parameters: one flat buffer (single all-reduce, single SGD launch) -----------------
    def build(self, f_in):
        h = self.hidden or f_in
        c = self.n_labels
        self.f_in, self.hidden = f_in, h
        shapes = {"w1": (f_in, h), "b1": (h,), "w2": (h, h), "b2": (h,), "w3": (h, c), "b3": (c,)}
        self.n_params = sum(int(np.prod(s)) for s in shapes.values())
        # grads carries two extra floats (loss sum, correct count) so that they ride in the
        # same all-reduce as the gradients.
        self.flat_p = self.ctx.zeros(self.n_params)
        self.flat_g = self.ctx.zeros(self.n_params + 2)
        self.p, self.g = {}, {}
        off = 0
        for k in self.PARAM_ORDER:
            n = int(np.prod(shapes[k]))
            self.p[k] = self.flat_p.flat(off, n, shapes[k])
            self.g[k] = self.flat_g.flat(off, n, shapes[k])
            off += n
        self.loss_acc = self.flat_g.flat(off, 2)
        init = {"w1": glorot_uniform(self._rng, f_in, h), "w2": glorot_uniform(self._rng, h, h),
                "w3": glorot_uniform(self._rng, h, c)}
        for k, v in init.items():
            self.p[k].copy_from_host(v)
        self.built = True

    def get_weights(self):
        return [self.p[k].numpy() for k in self.PARAM_ORDER]

    def set_weights(self, weights):
        for k, w in zip(self.PARAM_ORDER, weights):
            self.p[k].copy_from_host(np.asarray(w, np.float32).reshape(self.p[k].shape))

    @property
    def trainable_variables(self):
        return [self.p[k] for k in self.PARAM_ORDER]

    @property
    def losses(self):
        return []  # no regularisers (gcn.py:335 adds sum(model.losses))

    # ---- buffers for one batch shape ---------------------------------------------------------
    def _ensure(self, batch):
        if not self.built:
            self.build(batch.f)
        key = (batch.n, batch.n_graphs)
        if self._bufs is not None and self._bufs["key"] == key:
            return self._bufs
        ctx, n, b, h, c = self.ctx, batch.n, batch.n_graphs, self.hidden, self.n_labels
        self._drop_graphs()
        self._bufs = {
            "key": key,
            "h": ctx.empty((n, h)), "y1": ctx.empty((n, h)), "y2": ctx.empty((n, h)), "dz": ctx.empty((n, h)),
            "pooled": ctx.empty((b, h)), "logits": ctx.empty((b, c)), "probs": ctx.empty((b, c)),
            "dlogits": ctx.empty((b, c)), "dpooled": ctx.empty((b, h)),
            "arg": ctx.empty((b, h), np.int32) if self.pool == "max" else None,
        }
        return self._bufs

    def _drop_graphs(self):
        for g in self._graphs.values():
            if not isinstance(g, str):
                g.destroy()
        self._graphs = {}

    # ---- the call sequences --------------------------------------------------------------------
    def _forward(self, batch, bufs, with_loss, denom):
        ctx, p, prec = self.ctx, self.p, self.prec
        D.gemm(ctx, batch.x, p["w1"], None, bufs["h"], prec=prec)
        D.spmm(ctx, batch.a, bufs["h"], p["b1"], bufs["y1"], act="relu")
        D.gemm(ctx, bufs["y1"], p["w2"], None, bufs["h"], prec=prec)
        D.spmm(ctx, batch.a, bufs["h"], p["b2"], bufs["y2"], act="relu")
        D.segment_pool(ctx, batch.seg, bufs["y2"], bufs["pooled"], self.pool, bufs["arg"])
        D.gemm(ctx, bufs["pooled"], p["w3"], p["b3"], bufs["logits"], prec="f32")
        if with_loss:
            self.loss_acc.fill_zero()
            D.softmax_cce(ctx, bufs["logits"], batch.y, bufs["probs"], self.loss_acc, bufs["dlogits"], denom)

    def _backward(self, batch, bufs):
        ctx, p, g, prec = self.ctx, self.p, self.g, self.prec
        at = batch.a.transpose()
        D.gemm_dw(ctx, bufs["pooled"], bufs["dlogits"], g["w3"], prec="f32")
        D.act_bias_grad(ctx, bufs["dlogits"], None, bufs["dlogits"], None, db=g["b3"])
        D.gemm_dx(ctx, bufs["dlogits"], p["w3"], bufs["dpooled"], prec="f32")
        # pool gradient with the ReLU mask of layer 2 and its bias gradient fused
        D.segment_pool_bwd(ctx, batch.seg, bufs["dpooled"], bufs["dz"], self.pool, bufs["arg"], y=bufs["y2"], db=g["b2"])
        D.spmm(ctx, at, bufs["dz"], None, bufs["h"])                           # dH2 = A^T dZ2
        D.gemm_dw(ctx, bufs["y1"], bufs["h"], g["w2"], prec=prec)              # dW2 = Y1^T dH2
        D.gemm_dx(ctx, bufs["h"], p["w2"], bufs["dz"], prec=prec, y_mask=bufs["y1"], db=g["b1"])  # dZ1, db1
        D.spmm(ctx, at, bufs["dz"], None, bufs["h"])                           # dH1 = A^T dZ1
        D.gemm_dw(ctx, batch.x, bufs["h"], g["w1"], prec=prec)                 # dW1 = X^T dH1

    def _world(self):
        return self.comm.world_size if self.comm is not None else 1

    def _bind(self, batch):
        """Captured graphs hold the pointers of one batch: drop them when the batch changes."""
        if getattr(self, "_bound_uid", None) != batch.uid:
            for tag in [t for t in self._graphs if t[0] != "sgd"]:
                g = self._graphs.pop(tag)
                if not isinstance(g, str):
                    g.destroy()
            self._bound_uid = batch.uid

    def _run(self, tag, fn):
        """Run fn eagerly the first time (sizes the workspace), then capture + replay."""
        if not self.use_graph:
            fn()
            return
        st = self._graphs.get(tag)
        if st is None:
            fn()
            self._graphs[tag] = "warm"
        elif st == "warm":
            self._graphs[tag] = self.ctx.capture(fn)
            self._graphs[tag].launch()
        else:
            st.launch()

    # ---- public surface: model(inputs, training=...) and train_step ------------------------
    def _as_batch(self, inputs, target=None):
        if isinstance(inputs, DeviceBatch):
            if target is not None and inputs.y is None:
                inputs.y = self.ctx.to_device(target, np.float32)
            return inputs
        return DeviceBatch.from_host(self.ctx, inputs, target)

    def __call__(self, inputs, training=False):
        """model([x, a, i], training=False) -> probabilities [B, C] (gcn.py:351)."""
        batch = self._as_batch(inputs)
        bufs = self._ensure(batch)
        self._bind(batch)
        self._run(("fwd", batch.uid), lambda: (self._forward(batch, bufs, False, None),
                                              self._softmax_only(bufs)))
        return bufs["probs"].numpy()

    def _softmax_only(self, bufs):
        # probabilities without labels: y = zeros gives loss 0; reuse the fused kernel
        if "zero_y" not in bufs or bufs["zero_y"].shape != bufs["logits"].shape:
            bufs["zero_y"] = self.ctx.zeros(bufs["logits"].shape)
            bufs["scratch2"] = self.ctx.zeros(2)
        D.softmax_cce(self.ctx, bufs["logits"], bufs["zero_y"], bufs["probs"], bufs["scratch2"], None, None)

    def loss_and_grads(self, inputs, target, global_batch=None):
        """Forward + loss + every gradient (no update).  Returns (loss, acc)."""
        batch = self._as_batch(inputs, target)
        bufs = self._ensure(batch)
        denom = float(global_batch or batch.n_graphs)

        def seq():
            self._forward(batch, bufs, True, denom)
            self._backward(batch, bufs)
        self._bind(batch)
        self._run(("grad", batch.uid, denom), seq)
        if self.comm is not None and self.comm.world_size > 1:
            self.comm.allreduce_sum(self.flat_g)
        self._last_batch = batch
        return batch

    def train_step(self, inputs, target=None, lr=0.02, global_batch=None, fetch=True):
        """One optimisation step (gcn.py:330-340).  With a communicator the batch given here is
        this rank's shard and ``global_batch`` the number of graphs over all ranks."""
        batch = self.loss_and_grads(inputs, target, global_batch)
        self._run(("sgd", float(lr)), lambda: D.sgd(self.ctx, self.flat_p, self.flat_g.flat(0, self.n_params), lr))
        if not fetch:
            return None
        return self.fetch_metrics(global_batch or batch.n_graphs)

    def fetch_metrics(self, n_graphs):
        la = self.loss_acc.numpy()
        return float(la[0]), float(la[1]) / float(n_graphs)

    def evaluate_batch(self, inputs, target):
        """Forward + loss/acc only (the body of evaluate(), gcn.py:350-357)."""
        batch = self._as_batch(inputs, target)
        bufs = self._ensure(batch)
        self._forward(batch, bufs, True, float(batch.n_graphs))
        la = self.loss_acc.numpy()
        return float(la[0]), float(la[1]) / batch.n_graphs, bufs["probs"].numpy()

    def gradients(self):
        return {k: self.g[k].numpy() for k in self.PARAM_ORDER}
