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

import os

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
    def from_host(cls, ctx, inputs, y=None, normalize=None, weighted=True, symmetric=None):
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


class _Capacity:
    """Grow-only device storage behind per-batch views.  A streamed epoch brings a new (N, B) with every batch;
    re-allocating the activations per step (tens of MB) cost more than the step itself."""

    def __init__(self, ctx):
        self.ctx, self.store = ctx, {}

    def view(self, name, rows, width, dtype=np.float32, zero=False):
        need = int(rows) * int(width)
        cur = self.store.get(name)
        if cur is None or cur.size < need or cur.dtype != np.dtype(dtype):
            size = max(need, int(1.25 * cur.size) if cur is not None else 0, 1)
            cur = (self.ctx.zeros if zero else self.ctx.empty)(size, dtype)
            self.store[name] = cur
        return D.DeviceArray._view(cur, 0, (int(rows), int(width)))


class _GraphRunner:
    """Runs a fixed call sequence eagerly once (sizes the workspace), captures it into a HIP graph on the second
    use and replays it afterwards -- the role tf.function plays at gcn.py:328.  Graphs hold the pointers of one
    batch and are dropped when the batch changes."""

    use_graph = True

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

    # ---- metrics without a host round trip per step ---------------------------------------------------------------
    _MRING = 1024

    def stash_metrics(self, n_graphs):
        """After train_step(..., fetch=False): keep this step's (loss, hit count) on the device -- an 8-byte copy in
        stream order -- instead of reading it back; collect_metrics() returns the stashed steps.  A loop that reads
        loss and accuracy after every step waits for the GPU every step, and the GPU then waits for the host to queue
        the next one: the reference's loop only uses them at the end of the epoch (gcn.py:375-377)."""
        if getattr(self, "_mring", None) is None:
            self._mring, self._mcount, self._mdone = self.ctx.zeros(2 * self._MRING), [], []
        if len(self._mcount) == self._MRING:
            self._mdone.extend(self._read_ring())
        k = len(self._mcount)
        self.ctx._ck(self.ctx.lib.gcnx_d2d(self.ctx.h, self._mring.ptr + 8 * k, self.loss_acc.ptr, 8))
        self._mcount.append(float(n_graphs))

    def _read_ring(self):
        la = self._mring.numpy().reshape(-1, 2)[:len(self._mcount)]
        out = [(float(l), float(h) / n) for (l, h), n in zip(la, self._mcount)]
        self._mcount = []
        return out

    def collect_metrics(self):
        """[(loss, accuracy)] of the steps stashed since the last call (one device -> host copy)."""
        if getattr(self, "_mring", None) is None:
            return []
        out = self._mdone + self._read_ring()
        self._mdone = []
        return out

    # ---- the learning rate as a device scalar --------------------------------------------------------------------------
    # A rate passed by value is an argument of the captured update launch: one graph per VALUE (the reference's
    # PiecewiseConstantDecay has three, gcn.py:321-325; a schedule that moves every step would capture every step).  With the
    # rate in a device scalar (gcnx_set_lr_source) one captured step serves them all and a new value costs a 4-byte queued copy.
    lr_on_device = os.environ.get("GCNX_LR_DEVICE", "1") != "0"

    def _lr_key(self, lr):
        """Announce this step's rate; returns what the captured graphs are keyed on ("dev", or the value itself)."""
        if not self.lr_on_device or lr is None:
            self.ctx.set_lr_source(None)
            return lr
        if getattr(self, "_lr_buf", None) is None:
            self._lr_buf, self._lr_val = self.ctx.zeros(1), None
        self.ctx.set_lr_source(self._lr_buf)                # (a ctx can serve several models: each step names its own source)
        if self._lr_val != float(lr):
            self._lr_buf.copy_from_host(np.asarray([lr], np.float32), wait=False)
            self._lr_val = float(lr)
        return "dev"

    def _drop_graphs(self):
        for g in getattr(self, "_graphs", {}).values():
            if not isinstance(g, str):
                g.destroy()
        self._graphs = {}


class GCN2(_GraphRunner):
    PARAM_ORDER = ("w1", "b1", "w2", "b2", "w3", "b3")

    @D.with_default_context
    def __init__(self, ctx, n_labels=2, hidden=None, pool="sum", prec="f32", seed=0, comm=None, use_graph="auto",
                 cce_train="logits", cce_eval="probs"):
        """cce_train / cce_eval: which branch of keras.backend.categorical_crossentropy the loss follows (see
        gcnx_cce_mode in include/gcnx.h).  train_step runs under tf.function (gcn.py:328-335), where Keras swaps the
        Softmax op's output for its input and calls softmax_cross_entropy_with_logits ("logits": no clip, gradient
        (p - y)/B everywhere); evaluate() is eager (gcn.py:351-354) and, on TF < 2.6, takes the renormalise-and-clip
        branch ("probs").  cce_eval="logits" is the TF >= 2.6 behaviour (the output carries _keras_logits).  The
        reference pins neither TensorFlow nor Spektral, so both stay selectable: PARITY UNPINNED."""
        self.ctx, self.n_labels, self.hidden, self.pool, self.prec = ctx, int(n_labels), hidden, pool, prec
        self.cce_train, self.cce_eval = cce_train, cce_eval
        self.comm = comm                     # gcnx.comm.Communicator or None
        # use_graph: True -- every step replays ONE captured HIP graph (the tf.function of gcn.py:328); False -- eager launches;
        # "auto" (default, r4): eager where the step is the five launches of the one-launch layers (E. coli-sized batches, F <= 128,
        # binary labels, one process), captured everywhere else.  Back-to-back replays of a graph leave ~10 us between the last
        # kernel of one and the first of the next on this runtime; five eager launches per step keep the queue full from Python
        # and have no such boundary: config 2 0.1020 -> 0.0948 ms per step on the same box (the 12-to-88-launch steps of large
        # batches and of GeneralGNN are host-bound when issued eagerly: there the graph wins).
        self._graph_opt = use_graph
        self.use_graph = use_graph is not False
        # tuning knobs (diagnostics; DESIGN section 7: the knob list), read ONCE here -- the sequence a model runs never changes under it
        self._knob = {"fold": os.environ.get("GCNX_FOLD", "1") != "0", "duo": os.environ.get("GCNX_DUO", "1") != "0",
                      "fused": os.environ.get("GCNX_FUSED", "1") != "0", "side": int(os.environ.get("GCNX_SIDE", "1")),
                      "head_late": os.environ.get("GCNX_HEAD_LATE", "1") != "0",
                      "s_order": os.environ.get("GCNX_S_ORDER", "1") != "0",
                      "buckets": os.environ.get("GCNX_COMM_BUCKETS", "1") != "0",
                      "act16": os.environ.get("GCNX_ACT16", "1") != "0",
                      "pool_in_spmm": os.environ.get("GCNX_POOL_IN_SPMM", "1") != "0"}
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
        """Activation buffers for this batch shape.  Storage is grow-only capacity (a streamed epoch brings a new
        (N, B) with every batch: re-allocating ~60 MB per step cost more than the step); the returned arrays are
        views of exactly the batch's shape."""
        if not self.built:
            self.build(batch.f)
        key = (batch.n, batch.n_graphs)
        if self._bufs is not None and self._bufs["key"] == key:
            return self._bufs
        n, b, h, c = batch.n, batch.n_graphs, self.hidden, self.n_labels
        self._drop_graphs()
        if getattr(self, "_cap", None) is None:
            self._cap = _Capacity(self.ctx)
        v = self._cap.view
        self._bufs = {"key": key}
        for k in ("h", "y1", "y2", "dz", "h2", "dz2"):          # h2 / dz2: the side section still reads h / dz
            self._bufs[k] = v(k, n, h)
        if self._fused(batch):                                  # one-launch layers: S1 = A X, S2 = A Y1 (operands of dW)
            self._bufs["s1"], self._bufs["s2"] = v("s1", n, self.f_in), v("s2", n, h)
            self._bufs["w2t"] = v("w2t", h, h)                  # W2^T, a by-product of layer 2's forward launch
            # head inside the backward: the pool's per-tile partial sums / positive counts (layer 2's launch writes them)
            # and the per-graph totals (the backward launch does)
            tr = D.pool_tile_rows(n, b)
            self._bufs["tp_part"], self._bufs["tp_cnt"] = v("tp_part", tr, h), v("tp_cnt", tr, h)
            self._bufs["pool_sum"], self._bufs["pool_cnt"] = v("pool_sum", b, h), v("pool_cnt", b, h)
        elif self._s_order():
            self._bufs["s1"] = v("s1", n, self.f_in)            # S1 = A X: layer 1 evaluated as (A X) W1, dW1 = S1^T dZ1
        for k, w in (("pooled", h), ("probs", c), ("dpooled", h)):
            self._bufs[k] = v(k, b, w)
        self._bufs["arg"] = v("arg", b, h, np.int32) if self.pool == "max" else None
        return self._bufs

    # ---- the call sequences --------------------------------------------------------------------
    def _forward(self, batch, bufs, with_loss, denom):
        ctx, p, prec = self.ctx, self.p, self.prec
        if self._fused(batch):
            # small-feature regime: each GCNConv is one launch, evaluated as (A X) W (gcnx_gcn_conv_fwd); A X is kept
            # for the weight gradient when a backward pass follows
            keep = with_loss == "grads"
            late = keep and self._head_late(batch)
            D.gcn_conv_fwd(ctx, batch.a, batch.x, p["w1"], p["b1"], bufs["y1"], act="relu", s=bufs["s1"] if keep else None,
                           prec=prec)
            D.gcn_conv_fwd(ctx, batch.a, bufs["y1"], p["w2"], p["b2"], bufs["y2"], act="relu", s=bufs["s2"] if keep else None,
                           wt=bufs["w2t"] if keep else None, prec=prec,
                           pool=(batch.seg, bufs["tp_part"], bufs["tp_cnt"]) if late else None)
        else:
            bufs["act16"] = False
            if self._s_order() and self._act16_try(batch, with_loss):
                # bf16 STORAGE of what only bf16-operand weight GEMMs read (r3): with prec = "bf16" every weight GEMM rounds its
                # operands to bfloat16 as it loads them, so S1 = A X and Y1 (read by the two products of layer 2's kernel and
                # by the two weight-gradient products) are stored rounded -- the same bits in every result, half the bytes in
                # the aggregation's output, in both operands of dW1 / dW2 and in the input of three of the five GEMMs.
                s16 = self._cap.view("s1_16", batch.n, self.f_in, np.uint16)
                if D.spmm_bf16out(ctx, batch.a, batch.x, None, s16):          # (False: the tile kernels do not serve this batch)
                    y16 = self._cap.view("y1_16", batch.n, self.hidden, np.uint16)
                    bufs["y1bits"] = self._cap.view("y1bits", batch.n, 16, np.int32) if with_loss == "grads" else None
                    # the bf16 images of the step's three weight operands (W1 and W2 forward, W2 for dX) in one launch
                    bufs["wimg"] = D.stream_images(ctx, [(p["w1"], True), (p["w2"], True), (p["w2"], False)],
                                                   self._cap.view("wimg", 3, 65536, np.uint16))
                    if not (D.gemm_fwd_bf16(ctx, s16, p["w1"], p["b1"], y16, act="relu", bits=bufs["y1bits"], wimg=bufs["wimg"][0])
                            and D.gemm_fwd_bf16(ctx, y16, p["w2"], None, bufs["h"], wimg=bufs["wimg"][1])):
                        raise RuntimeError("gcnx: the streaming bf16 GEMM refused a shape the bf16-storage path was chosen for")
                    bufs["act16"], bufs["y1bits_ok"], bufs["s1_16"], bufs["y1_16"] = True, with_loss == "grads", s16, y16
            if bufs["act16"]:
                pass
            elif self._s_order():
                # layer 1 as (A X) W1: the same product as GCNConv's A (X W1), two launches either way -- but the layer's
                # input needs no gradient, so with S1 = A X kept dW1 = S1^T dZ1 and the backward pass has no aggregation
                # for this layer (one of the step's four, 575 us of 4.7 ms at config 3)
                D.spmm(ctx, batch.a, batch.x, None, bufs["s1"])
                # with a backward pass to follow and the streaming bf16 kernel serving the product: [Y1 > 0] also as a bit
                # image, which the dX launch of the backward pass reads instead of Y1 (1 GB -> 32 MB at config 3)
                bufs["y1bits_ok"] = False
                if (with_loss == "grads" and prec in ("bf16", "bf16x3") and self.hidden == 256 and self.f_in == 256
                        and batch.n >= 32768):            # (the streaming kernel's shape: K = 256 too)
                    bufs["y1bits"] = self._cap.view("y1bits", batch.n, 16, np.int32)     # grow-only storage, stable pointers
                    bufs["y1bits_ok"] = D.gemm_relu_bits(ctx, bufs["s1"], p["w1"], p["b1"], bufs["y1"], bufs["y1bits"], prec=prec)
                if not bufs["y1bits_ok"]:
                    D.gemm(ctx, bufs["s1"], p["w1"], p["b1"], bufs["y1"], act="relu", prec=prec)
            else:
                D.gemm(ctx, batch.x, p["w1"], None, bufs["h"], prec=prec)
                D.spmm(ctx, batch.a, bufs["h"], p["b1"], bufs["y1"], act="relu")
            if not bufs["act16"]:
                D.gemm(ctx, bufs["y1"], p["w2"], None, bufs["h"], prec=prec)
            # pooled layer on the tile kernels with a backward pass to follow: its launch also writes [Y2 > 0] as a bit
            # image, which the folded backward aggregation expands instead of reading Y2 again (1 GB -> 32 MB at config 3)
            bufs["y2bits_ok"] = bufs["pool_done"] = False
            if (self._fold(batch) and batch.a.plan is not None and self.hidden % 32 == 0
                    and (with_loss == "grads" or self._knob["pool_in_spmm"])):                 # (forward-only passes too: evaluate())
                bufs["y2bits"] = self._cap.view("y2bits", batch.n, self.hidden // 32, np.int32)
                # ... and on tile graphs the launch leaves the pooled rows and positive counts too: Y2 is neither written
                # nor read back by a pool launch there (r3; rows of the graphs taller than a tile are, as before)
                if self._knob["pool_in_spmm"] and not self._head_late(batch):
                    bufs["pool_cnt2"] = self._cap.view("pool_cnt2", batch.n_graphs, self.hidden)
                    bufs["pool_done"] = D.spmm_relu_bits_pool(ctx, batch.a, bufs["h"], p["b2"], bufs["y2"], bufs["y2bits"], batch.seg,
                                                              bufs["pooled"], bufs["pool_cnt2"], self.pool)
                bufs["y2bits_ok"] = bufs["pool_done"] or (with_loss == "grads" and
                                                          D.spmm_relu_bits(ctx, batch.a, bufs["h"], p["b2"], bufs["y2"], bufs["y2bits"]))
            if not bufs["y2bits_ok"]:
                D.spmm(ctx, batch.a, bufs["h"], p["b2"], bufs["y2"], act="relu")
        # Global pool, then Dense(softmax) + CCE + accuracy + the head gradients in one launch; with few graphs the
        # head also combines the pool's row-slice partial sums (gcnx_pool_dense_softmax_cce)
        head = dict(mode=self.pool, argmax=bufs["arg"])
        bufs["_head_late"] = None
        if with_loss == "grads" and self._head_late(batch):
            # one-launch layers with a backward pass to follow: neither a pool nor a head launch here.  Layer 2's launch
            # left the pool's per-tile partial sums; the backward aggregation adds them up and evaluates dPooled per graph
            # itself, and the rest of the head (probabilities, loss, accuracy, dW3, db3, db2 -- leaves) rides in the
            # weight-gradient launch: the pool's 7 us and the head's 10 are off the critical path.
            bufs["_head_late"] = D.head_args(batch.seg, bufs["tp_part"], bufs["tp_cnt"], bufs["pool_sum"], bufs["pool_cnt"], p["w3"],
                                             p["b3"], batch.y, denom, bufs["probs"], self.loss_acc, self.g["w3"], self.g["b3"],
                                             self.g["b2"], bufs["pooled"], bufs["dpooled"], mode=self.pool, cce=self.cce_train)
        elif with_loss == "grads" and bufs.get("pool_done"):
            D.pooled_dense_softmax_cce(ctx, batch.seg, bufs["pooled"], bufs["pool_cnt2"], p["w3"], p["b3"], batch.y, bufs["probs"],
                                       self.loss_acc, denom, dw=self.g["w3"], db=self.g["b3"], dpooled=bufs["dpooled"], mode=self.pool,
                                       db_relu=self.g["b2"], cce=self.cce_train)
        elif with_loss == "grads":
            # db2 rides along when the backward folds pool' into the aggregation (dZ2 is never materialised there)
            D.pool_dense_softmax_cce(ctx, batch.seg, bufs["y2"], bufs["pooled"], p["w3"], p["b3"], batch.y, bufs["probs"],
                                     self.loss_acc, denom, dw=self.g["w3"], db=self.g["b3"], dpooled=bufs["dpooled"],
                                     db_relu=self.g["b2"] if self._fold(batch) else None, cce=self.cce_train, **head)
        elif bufs.get("pool_done"):       # evaluate() / a plain forward: the head on the pooled rows the aggregation left
            D.pooled_dense_softmax_cce(ctx, batch.seg, bufs["pooled"], None, p["w3"], p["b3"], batch.y if with_loss else None, bufs["probs"],
                                       self.loss_acc if with_loss else None, denom, mode=self.pool, cce=self.cce_eval)
        elif with_loss:
            D.pool_dense_softmax_cce(ctx, batch.seg, bufs["y2"], bufs["pooled"], p["w3"], p["b3"], batch.y, bufs["probs"],
                                     self.loss_acc, denom, cce=self.cce_eval, **head)
        else:
            D.pool_dense_softmax_cce(ctx, batch.seg, bufs["y2"], bufs["pooled"], p["w3"], p["b3"], None, bufs["probs"], **head)

    def _backward(self, batch, bufs, lr=None, buckets=False):
        """All gradients; with ``lr`` (single process) the SGD step too -- returns True if it was applied here.
        buckets (multi-GPU): the gradient all-reduce is issued from here in two buckets where the sequence allows it
        (the two-launch layers of large batches); ``self._reduced_in_backward`` tells the caller."""
        ctx, p, g, prec = self.ctx, self.p, self.g, self.prec
        self._reduced_in_backward = False
        at = batch.a.transpose()
        # The gradient leaves of layer 2 (db2, dW2: nothing later in the backward pass reads them) run in ONE side
        # section, concurrently with the main chain dX -> db1 -> A^T -> dW1.  One fork and one join per step: each
        # costs ~10 us of cross-queue signalling, which is why there are not three sections.  The buffers the side
        # section reads (dz, h) stay unmodified until the join: the main chain continues in dz2 / h2.
        side = self._knob["side"]                      # tuning knob GCNX_SIDE: 0 = serial, 1 = one section, 7 = three sections
        if side != 1:
            return self._backward_knob(batch, bufs, 0 if side == 0 else 7)
        # SUM / AVG pooling: dZ2 = pool'(dPooled) * [Y2 > 0] is never materialised -- the aggregation gathers the mask
        # from Y2 (row gather in the latency regime, masked LDS tiles with a plan) and scales by the graph's dPooled
        # vector; db2 counts the mask (in the head with few graphs, gcnx_pool_bwd_colsum on the side stream otherwise).
        fold = self._fold(batch)
        if self._fused(batch):
            # pool' + ReLU' + A^T + W2^T + ReLU' in one launch (dZ2 and dZ1 out, db1 partials pending; db2 came out of
            # the head), then both weight gradients -- dW1 = S1^T dZ1, dW2 = S2^T dZ2 -- and the update in the last two
            ha = bufs.get("_head_late")
            pend = D.gcn_conv_bwd_pool(ctx, at, bufs["y2"], batch.seg, None if ha is not None else bufs["dpooled"], p["w2"], bufs["y1"],
                                       bufs["dz"], bufs["dz2"], db1=g["b1"], mode=self.pool, scratch=self._defer_scratch(batch),
                                       w2t=bufs["w2t"], prec=prec, head=ha)
            if lr is None:
                D.gemm_dw2(ctx, bufs["s1"], bufs["dz2"], g["w1"], bufs["s2"], bufs["dz"], g["w2"], prec="f32",
                           grads=self.flat_g.flat(0, self.n_params), pending=pend, leaf=ha)
                return False
            D.gemm_dw2(ctx, bufs["s1"], bufs["dz2"], g["w1"], bufs["s2"], bufs["dz"], g["w2"], prec="f32", params=self.flat_p,
                       grads=self.flat_g.flat(0, self.n_params), lr=lr, pending=pend, leaf=ha)
            return True
        act16 = bool(bufs.get("act16"))
        dh16 = self._cap.view("dh2_16", batch.n, self.hidden, np.uint16) if act16 else None
        dh16_done = False
        if fold and act16 and bufs.get("y2bits_ok"):
            dh16_done = D.spmm_pool_bwd_bf16out(ctx, at, bufs["y2"], batch.seg, bufs["dpooled"], dh16, self.pool, y_bits=bufs["y2bits"])
        if dh16_done:
            pass                                                                # dH2 = A^T dZ2, stored as bf16
        elif fold:
            D.spmm_pool_bwd(ctx, at, bufs["y2"], batch.seg, bufs["dpooled"], bufs["h"], self.pool,
                            y_bits=bufs["y2bits"] if bufs.get("y2bits_ok") else None)                # dH2 = A^T dZ2
        else:
            D.segment_pool_bwd(ctx, batch.seg, bufs["dpooled"], bufs["dz"], self.pool, bufs["arg"], y=bufs["y2"])  # dZ2 (ReLU mask fused)
            D.spmm(ctx, at, bufs["dz"], None, bufs["h"])                       # dH2 = A^T dZ2
        if fold and batch.a.plan is None and self._knob["duo"]:
            # db2 came out of the head, and dW2 shares dX's launch (gcnx_dense_bwd): one stream, no fork / join --
            # the second stream's signalling cost 25 us of a 194 us step
            if lr is None:
                D.dense_bwd(ctx, bufs["y1"], bufs["h"], p["w2"], bufs["dz2"], g["w2"], prec=prec, y_mask=bufs["y1"],
                            db_prev=g["b1"])                                   # dW2, dZ1, db1
                if buckets:
                    self._allreduce_tail_bucket()                              # beside layer 1's dW (and its aggregation)
                xs, dh1 = self._layer1_dw_operands(batch, bufs, at)
                D.gemm_dw(ctx, xs, dh1, g["w1"], prec=prec)                    # dW1 = X^T (A^T dZ1) or S1^T dZ1
                if buckets:
                    self._allreduce_head_bucket()
                return False
            # With the update in the same step, the reductions that finish the leaves dW2 / db1 wait for the last
            # launch: the split-K reduction of dW1 folds them in and applies the SGD step to every parameter.
            pend = D.dense_bwd_deferred(ctx, bufs["y1"], bufs["h"], p["w2"], bufs["dz2"], g["w2"], self._defer_scratch(batch),
                                        prec=prec, y_mask=bufs["y1"], db_prev=g["b1"])
            xs, dh1 = self._layer1_dw_operands(batch, bufs, at)
            D.gemm_dw_sgd(ctx, xs, dh1, g["w1"], self.flat_p, self.flat_g.flat(0, self.n_params), lr, prec=prec,
                          pending=pend)
            return True
        # The side section pays only while the main chain still has layer 1's backward aggregation to run beside the
        # leaves.  With layer 1 in (A X) W1 order the chain is dX -> dW1, streaming GEMMs that share one bound (HBM, or
        # the fp32 MFMA) with dW2: on one stream the step measured 1-3 % faster at config 3 in every precision (4.12
        # against 4.21-4.25 ms in bf16), the same at config 5.
        import contextlib
        if act16 and not dh16_done:
            D.to_bf16_into(ctx, bufs["h"], dh16)                                # (dH2 came from an fp32 path: the same rounding)
        with (contextlib.nullcontext() if "s1" in bufs else ctx.side()):
            if not fold:
                D.act_bias_grad(ctx, bufs["dz"], None, bufs["dz"], None, db=g["b2"])     # db2 = colsum(dZ2)
            # (folded: db2 came out of the head -- from the pool's own count of positive entries)
            if act16:
                self._must(D.gemm_dw_bf16(ctx, bufs["y1_16"], dh16, g["w2"]))   # dW2 = Y1^T dH2
            else:
                D.gemm_dw(ctx, bufs["y1"], bufs["h"], g["w2"], prec=prec)      # dW2 = Y1^T dH2
        if buckets:
            # Multi-GPU ("enqueue behind the layer-1 dW GEMM"): the gradients of layers 2 and 3
            # and the metric tail -- {dW2, db2, dW3, db3, loss, #correct}, the contiguous tail of the flat buffer -- are
            # final here, so their all-reduce runs on the side stream while layer 1's backward (dX, dW1: two of the step's
            # five weight GEMMs) runs on the main one; {dW1, db1} follows as a second bucket behind the join (collectives of
            # one communicator never overlap each other).  A transport that cannot be captured (the thread-rank test
            # communicator reduces through the host) issues the same two buckets in line.
            ctx.join()
            self._allreduce_tail_bucket()
        if act16:
            dz16 = self._cap.view("dz1_16", batch.n, self.hidden, np.uint16)
            self._must(D.gemm_dx_bf16(ctx, dh16, p["w2"], dz16, mask_bits=bufs["y1bits"], db=g["b1"], wimg=bufs["wimg"][2]))   # dZ1 (bf16), db1
            self._must(D.gemm_dw_bf16(ctx, bufs["s1_16"], dz16, g["w1"]))      # dW1 = S1^T dZ1
        else:
            D.gemm_dx(ctx, bufs["h"], p["w2"], bufs["dz2"], prec=prec, y_mask=bufs["y1"], db=g["b1"],
                      mask_bits=bufs["y1bits"] if bufs.get("y1bits_ok") else None)                   # dZ1, db1
            xs, dh1 = self._layer1_dw_operands(batch, bufs, at)
            D.gemm_dw(ctx, xs, dh1, g["w1"], prec=prec)                        # dW1 = X^T (A^T dZ1) or S1^T dZ1
        ctx.join()
        if buckets:
            self._allreduce_head_bucket()

    @staticmethod
    def _must(ok):
        if not ok:
            raise RuntimeError("gcnx: a bf16-storage kernel refused a shape its producer had accepted")

    def _act16_try(self, batch, with_loss):
        """bf16 storage of S1, Y1, dH2, dZ1 (the tensors only weight GEMMs read): a step -- training, or the forward pass of
        evaluate() -- with plain bf16 GEMM operands at the streaming kernels' shape, the aggregation on a tile plan.  The first producer (gcnx_spmm_csr_bf16out) has the
        last word.  GCNX_ACT16=0 keeps fp32 storage (same results, bit for bit: tested)."""
        return (self.prec == "bf16" and self._knob["act16"] and self.hidden == 256 and self.f_in == 256
                and batch.n >= 32768 and batch.a.plan is not None and batch.a.vals is not None and self._knob["side"] == 1)

    def _allreduce_tail_bucket(self):
        """{dW2, db2, dW3, db3, loss, #correct}: on the side stream (a second branch of a captured step) when the transport
        enqueues on a stream; in line otherwise."""
        import contextlib
        off = self._bucket_split()
        tail = self.flat_g.flat(off, self.n_params + 2 - off)
        with (self.ctx.side() if getattr(self.comm, "capturable", False) else contextlib.nullcontext()):
            self.comm.allreduce_sum(tail)

    def _allreduce_head_bucket(self):
        """{dW1, db1}, behind the join: collectives of one communicator never overlap each other."""
        self.ctx.join()
        self.comm.allreduce_sum(self.flat_g.flat(0, self._bucket_split()))
        self._reduced_in_backward = True

    def _bucket_split(self):
        """First element of the all-reduce's second-layer bucket in the flat gradient buffer: [w1, b1 | w2, b2, w3, b3, loss, acc]."""
        return int(np.prod(self.p["w1"].shape)) + int(np.prod(self.p["b1"].shape))

    def _s_order(self):
        """Layer 1 as (A X) W1 in the two-launch (non-fused) paths: when the aggregation is not wider that way (F <= H)."""
        return self._knob["s_order"] and self.built and self.f_in <= self.hidden

    def _layer1_dw_operands(self, batch, bufs, at):
        """The two operands of dW1 with dZ1 in bufs["dz2"]: (S1, dZ1) when the forward kept S1 = A X, otherwise
        (X, A^T dZ1) -- after the backward aggregation of layer 1."""
        if "s1" in bufs and not self._fused(batch):
            return bufs["s1"], bufs["dz2"]
        D.spmm(self.ctx, at, bufs["dz2"], None, bufs["h2"])                    # dH1 = A^T dZ1
        return batch.x, bufs["h2"]

    def _defer_scratch(self, batch):
        """Device buffer that holds the deferred partial results of layer 2's dense backward (grow-only)."""
        need = max(D.dense_bwd_scratch_floats(self.ctx, batch.n, self.hidden, self.hidden),
                   D.gcn_conv_bwd_scratch_floats(self.ctx, batch.n, self.hidden))
        cur = getattr(self, "_defer_buf", None)
        if cur is None or cur.size < need:
            self._defer_buf = self.ctx.empty(max(need, 4))
        return self._defer_buf

    def _fold(self, batch):
        """SUM / AVG pooling: pool' and the ReLU mask fold into the backward aggregation (dZ2 is never materialised) and
        db2 = sum_g dPooled[g] * #[Y2_g > 0] comes out of the head (few graphs) or of gcnx_pool_bwd_colsum."""
        return (self.pool in ("sum", "avg") and self.hidden % 4 == 0 and (batch.a.plan is None or self.hidden % 32 == 0)
                and self._knob["fold"] and self._knob["side"] == 1)

    def _fused(self, batch):
        """The small-feature regime (config 2): every GCNConv and the backward from the pool down to dZ1 are single
        launches (csrc/fused.hip).  Needs the folded backward's conditions and F, H in {32, 64, 128}; prec "f32" (exact
        fp32 products) or "bf16x3" (split-bf16 products in the conv launches; the two weight gradients stay on the fp32
        MFMA -- sums over N rows, at least as accurate)."""
        route = batch.__dict__.setdefault("_route", {})      # asked several times per step: once per (batch, model)
        key = ("fused", id(self), self.prec, self.built, batch.a.plan is None)
        if key not in route:
            route[key] = bool(self.built and self._fold(batch) and self.prec in ("f32", "bf16x3") and self._knob["fused"]
                              and self._knob["duo"] and self.f_in in (32, 64, 128) and self.hidden in (32, 64, 128)
                              and D.gcn_conv_fused_ok(self.ctx, batch.n, self.f_in, self.hidden)
                              and D.gcn_conv_fused_ok(self.ctx, batch.n, self.hidden, self.hidden))
        return route[key]

    def _head_late(self, batch):
        """One-launch layers (see _fused; SUM / AVG pooling), at most 2 classes (the reference's binary labels) and no graph
        without nodes: pool and classifier head are evaluated inside the forward / backward launches (gcnx_gcn_conv_fwd_pool,
        gcnx_head_args) instead of as launches between them."""
        return (self._knob["head_late"] and self._fused(batch) and self.n_labels <= 2 and batch.y is not None
                and not batch.seg.has_empty)

    def _backward_knob(self, batch, bufs, side):
        """The same backward with individual side sections switched off (GCNX_SIDE bits; measurement only)."""
        import contextlib
        ctx, p, g, prec = self.ctx, self.p, self.g, self.prec
        at = batch.a.transpose()
        sec = lambda on: ctx.side() if on else contextlib.nullcontext()
        D.segment_pool_bwd(ctx, batch.seg, bufs["dpooled"], bufs["dz"], self.pool, bufs["arg"], y=bufs["y2"])
        with sec(side & 1):
            D.act_bias_grad(ctx, bufs["dz"], None, bufs["dz"], None, db=g["b2"])
        D.spmm(ctx, at, bufs["dz"], None, bufs["h"])
        ctx.join()
        with sec(side & 2):
            D.gemm_dw(ctx, bufs["y1"], bufs["h"], g["w2"], prec=prec)
        D.gemm_dx(ctx, bufs["h"], p["w2"], bufs["dz"], prec=prec, y_mask=bufs["y1"])
        ctx.join()
        with sec(side & 4):
            D.act_bias_grad(ctx, bufs["dz"], None, bufs["dz"], None, db=g["b1"])
        D.spmm(ctx, at, bufs["dz"], None, bufs["h"])
        D.gemm_dw(ctx, batch.x, bufs["h"], g["w1"], prec=prec)
        ctx.join()

    def _world(self):
        return self.comm.world_size if self.comm is not None else 1

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
        self._run(("fwd", batch.uid), lambda: self._forward(batch, bufs, False, None))
        return bufs["probs"].numpy()

    def _comm_in_graph(self):
        """Multi-GPU: the gradient all-reduce is recorded INTO the step graph (RCCL enqueues on the ctx stream and
        supports stream capture), so a sharded step is one graph launch -- gradient kernels, ncclAllReduce, SGD --
        instead of graph | eager collective | graph (two graph boundaries around a 0.14 ms step).  Falls back to
        that three-part form if the communicator cannot be captured (the thread-rank test transport) or a capture
        with the collective inside ever fails."""
        return (self.use_graph and self._world() > 1 and getattr(self.comm, "capturable", False)
                and not getattr(self, "_comm_capture_failed", False))

    def loss_and_grads(self, inputs, target, global_batch=None, _lr=None):
        """Forward + loss + every gradient (no update unless ``_lr``).  Returns the device batch."""
        batch = self._as_batch(inputs, target)
        bufs = self._ensure(batch)
        denom = float(global_batch or batch.n_graphs)
        multi = self._world() > 1
        if self._graph_opt == "auto":
            self.use_graph = bool(multi or not (self._fused(batch) and self._head_late(batch)))
        fused_comm = multi and _lr is not None and self._comm_in_graph()

        # (the bucketed form needs the side stream inside the step: with a capturable communicator only when the collective
        # is recorded into the step graph anyway, or the step runs eagerly)
        buckets = multi and self._knob["buckets"] and (fused_comm or not self.use_graph)

        def seq():
            self._forward(batch, bufs, "grads", denom)
            if self._backward(batch, bufs, None if multi else _lr, buckets=buckets):
                return
            if fused_comm and not self._reduced_in_backward:
                self.comm.allreduce_sum(self.flat_g)
            if _lr is not None and (fused_comm or not multi):
                # the update rides in the same captured graph (one graph launch per step)
                D.sgd(self.ctx, self.flat_p, self.flat_g.flat(0, self.n_params), _lr)
        self._bind(batch)
        self._step_applied = fused_comm or not multi
        lr_key = self._lr_key(_lr)
        try:
            self._run(("grad", batch.uid, denom, lr_key if self._step_applied else None), seq)
        except Exception as e:
            if not fused_comm:
                raise
            # capture with the collective inside failed: from now on graph | all-reduce | graph (every rank runs the
            # same software, so every rank lands here together)
            import sys
            print(f"gcnx: RCCL all-reduce could not be captured into the step graph ({e}); falling back to an eager "
                  f"collective between two graphs", file=sys.stderr)
            self._comm_capture_failed = True
            self._drop_graphs()
            return self.loss_and_grads(batch, None, global_batch, _lr)
        if multi and not fused_comm and not self._reduced_in_backward:
            self.comm.allreduce_sum(self.flat_g)
        self._last_batch = batch
        self.ctx.set_lr_source(None)                      # (eager gcnx_sgd calls of other users of this ctx take their argument)
        return batch

    def train_step(self, inputs, target=None, lr=0.02, global_batch=None, fetch=True):
        """One optimisation step (gcn.py:330-340).  With a communicator the batch given here is
        this rank's shard and ``global_batch`` the number of graphs over all ranks."""
        batch = self.loss_and_grads(inputs, target, global_batch, _lr=float(lr))
        if not self._step_applied:
            self._run(("sgd", self._lr_key(float(lr))), lambda: D.sgd(self.ctx, self.flat_p, self.flat_g.flat(0, self.n_params), lr))
            self.ctx.set_lr_source(None)
        if fetch == "stash":
            self.stash_metrics(global_batch or batch.n_graphs)
            return None
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


def evaluate(model, loader, normalize=None):
    """The reference's evaluate(loader) (src/scripts/gcn.py:342-362).  One implementation: gcnx.train.evaluate."""
    from .train import evaluate as _evaluate
    return _evaluate(model, loader, normalize)


class GeneralGNN(_GraphRunner):
    """spektral.models.GeneralGNN -- the model the reference trains (src/scripts/gcn.py:320:
    ``GeneralGNN(dataset.n_labels, activation="softmax")``), SURVEY 8.A.3:

        pre-MLP(2) -> 4 x [z = GeneralConv(out); out = concat([z, out])] -> GlobalSumPool -> post-MLP(2)

    MLP layer = Dense -> BatchNormalization -> Dropout(0) -> PReLU (last post layer: softmax);
    GeneralConv = Dense -> BN -> PReLU -> sum-aggregation over a.indices (values ignored).
    Same constructor signature (activation "softmax" as gcn.py:320 passes it, or Spektral's default None = linear head); of
    the other options those listed in __init__ are built, anything else raises.  The skip concatenation is never materialised: every layer reads / writes a
    column slice of one [N, hidden*(message_passing+1)] buffer through leading-dimension views.
    Weights are exposed in Keras order per layer: kernel, bias, gamma, beta, moving_mean,
    moving_variance, alpha.  With a communicator (one process per GPU, a graph shard each) BatchNormalization is
    synchronised: the column sums of both moment passes and of the backward pass are all-reduced, so every rank
    normalises with the statistics of the GLOBAL batch, and the step equals the single-GPU step on the whole batch.
    """

    @D.with_default_context
    def __init__(self, ctx, output, activation=None, hidden=256, message_passing=4, pre_process=2, post_process=2,
                 connectivity="cat", batch_norm=True, dropout=0.0, aggregate="sum", hidden_activation="prelu", pool="sum",
                 prec="f32", seed=0, use_graph=True, comm=None, cce_train="logits", cce_eval="probs"):
        self.use_graph, self._graphs = use_graph, {}
        self.cce_train, self.cce_eval = cce_train, cce_eval     # see GCN2.__init__
        self.comm = comm                                  # gcnx.comm.Communicator: sync-BN + gradient all-reduce
        # activation: "softmax" (gcn.py:320) or None / "linear" (Spektral's own default: the last post layer ends with its
        # BatchNormalization, model(inputs) returns those logits).  A linear head is trained on the from-logits cross-entropy
        # -- CategoricalCrossentropy(from_logits=True), the only form defined on unnormalised outputs -- which is also what
        # tf.function makes of the softmax head (see GCN2.__init__), so the training step is the same launch sequence and
        # only what the model RETURNS differs.
        if activation == "linear":
            activation = None
        if activation not in ("softmax", None):
            raise NotImplementedError(f"GeneralGNN(activation={activation!r}): 'softmax' (gcn.py:320) and None are built")
        self.activation = activation
        if activation is None:
            self.cce_train = self.cce_eval = "logits"
        # Spektral's other options that map onto kernels that exist (r3; PARITY UNPINNED like the rest, the
        # oracle's restatement of them is checked against torch autograd): connectivity "sum" (out = z + out), batch_norm False
        # (no BatchNormalization layers), hidden_activation "relu" / None, dropout > 0 (the Dropout layer between
        # BatchNormalization and the activation of every MLP / GeneralConv layer; training only; this library's own generator)
        if connectivity not in ("cat", "sum"):
            raise NotImplementedError(f"GeneralGNN(connectivity={connectivity!r}): 'cat' (gcn.py:320) and 'sum' are built")
        if hidden_activation == "linear":
            hidden_activation = None
        if hidden_activation not in ("prelu", "relu", None):
            raise NotImplementedError(f"GeneralGNN(hidden_activation={hidden_activation!r}): 'prelu' (gcn.py:320), 'relu' and None are built")
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"GeneralGNN(dropout={dropout!r}): a rate in [0, 1)")
        self.connectivity, self.batch_norm, self.dropout = connectivity, bool(batch_norm), float(dropout)
        self.hidden_activation, self.seed = hidden_activation, int(seed)
        # Spektral's other aggregations / pools that map onto kernels that exist (r3): aggregate "mean" is the same gather with
        # the weight 1 / (entries of the row) per entry, pool "avg" / "max" are modes of the segment pool and its backward
        # ("prod", r4: tf.math.unsorted_segment_prod and its zero-aware gradient -- every aggregation Spektral names is built)
        if aggregate not in ("sum", "mean", "max", "min", "prod"):
            raise ValueError(f"GeneralGNN(aggregate={aggregate!r}): Spektral's aggregations are 'sum' (gcn.py:320), 'mean', 'max', 'min', 'prod'")
        if pool not in ("sum", "avg", "max"):
            raise NotImplementedError(f"GeneralGNN(pool={pool!r}): 'sum' (gcn.py:320), 'avg' and 'max' are built")
        self.aggregate, self.pool = aggregate, pool
        self.ctx, self.output, self.hidden, self.mp = ctx, int(output), int(hidden), int(message_passing)
        self.n_pre, self.n_post, self.prec = int(pre_process), int(post_process), prec
        self._rng = np.random.default_rng(seed)
        self.built = False
        self._bufs = None
        self._images_fresh, self._img_jobs = False, []

    # ---- parameters ------------------------------------------------------------------------------
    def build(self, f_in):
        h, mp = self.hidden, self.mp
        dims = []                                         # (group, fan_in, fan_out, has_prelu)
        w = f_in
        for _ in range(self.n_pre):
            dims.append(("pre", w, h, True)); w = h
        cat, bn, prelu = self.connectivity == "cat", self.batch_norm, self.hidden_activation == "prelu"
        for k in range(mp):
            dims.append(("gnn", h * (k + 1) if cat else h, h, True))
        w = self._wpool = h * (mp + 1) if cat else h
        for k in range(self.n_post):
            last = k == self.n_post - 1
            dims.append(("post", w, self.output if last else h, not last)); w = h
        n_train = sum(fi * fo + fo + (2 * fo if bn else 0) + (fo if pr and prelu else 0) for _, fi, fo, pr in dims)
        n_state = sum(2 * fo for _, _, fo, _ in dims) if bn else 0
        self.n_params = n_train
        ctx = self.ctx
        self.flat_p, self.flat_g = ctx.zeros(n_train), ctx.zeros(n_train + 2)
        self.flat_s = ctx.zeros(max(n_state, 1))          # moving_mean | moving_var per layer
        self.loss_acc = self.flat_g.flat(n_train, 2)
        self.layers = []
        self._step_dev = ctx.zeros(1, np.int32)           # the optimizer's step count on the device (Dropout streams)
        off = soff = 0
        for grp, fi, fo, pr in dims:
            L = {"group": grp, "fi": fi, "fo": fo, "act": self.hidden_activation if pr else None, "index": len(self.layers)}
            for name, shape in (("kernel", (fi, fo)), ("bias", (fo,))) + ((("gamma", (fo,)), ("beta", (fo,))) if bn else ()) + \
                    ((("alpha", (fo,)),) if pr and prelu else ()):
                n = int(np.prod(shape))
                L[name] = self.flat_p.flat(off, n, shape)
                L["g_" + name] = self.flat_g.flat(off, n, shape)
                off += n
            L["sums"], L["scratch"] = ctx.zeros(2 * fo), ctx.zeros(3 * fo)
            from .layers import glorot_uniform
            L["kernel"].copy_from_host(glorot_uniform(self._rng, fi, fo))
            if bn:
                L["moving_mean"] = self.flat_s.flat(soff, fo); L["moving_var"] = self.flat_s.flat(soff + fo, fo); soff += 2 * fo
                L["mean"], L["inv"] = ctx.zeros(fo), ctx.zeros(fo)
                L["gamma"].copy_from_host(np.ones(fo, np.float32))
                L["moving_var"].copy_from_host(np.ones(fo, np.float32))
                L["bn_gamma"], L["bn_beta"], L["bn_g_gamma"], L["bn_g_beta"] = L["gamma"], L["beta"], L["g_gamma"], L["g_beta"]
            else:
                # batch_norm=False: the fused batch-norm + activation passes run with the identity transform (mean 0, 1 / sigma 1,
                # gamma 1, beta 0, inference-mode backward): exactly act(z) and dy * act'(z); nothing of it is a parameter
                L["mean"], L["inv"] = ctx.zeros(fo), ctx.to_device(np.ones(fo, np.float32))
                L["bn_gamma"], L["bn_beta"] = L["inv"], L["mean"]
                L["bn_g_gamma"], L["bn_g_beta"] = ctx.zeros(fo), ctx.zeros(fo)
            self.layers.append(L)
        self.f_in, self.built = f_in, True
        self._alloc_images()

    def _alloc_images(self):
        """bf16 / bf16x3: weight images (MFMA-fragment order, csrc/gemm_panel.hip) of every Dense layer that runs on node
        rows -- the operand of X W and, where the layer's input needs a gradient, of dH W^T.  One launch per step rewrites
        them all (the weights change with every update)."""
        self._img_jobs = []
        self._img_prec = self.prec
        if self.prec == "f32":
            return
        ctx = self.ctx
        for i, L in enumerate(self.layers):
            L.pop("img_fwd", None); L.pop("img_bwd", None)
            if L["group"] == "post" or L["fo"] % 16 or L["fi"] % 4:
                continue
            ef = D.wimage_elems(ctx, L["fi"], L["fo"], False, self.prec)
            L["img_fwd"] = ctx.empty(ef, np.uint16)
            self._img_jobs.append((L["kernel"], L["img_fwd"], False, self.prec))
            if i > 0 and L["fi"] % 16 == 0:
                L["img_bwd"] = ctx.empty(D.wimage_elems(ctx, L["fi"], L["fo"], True, self.prec), np.uint16)
                self._img_jobs.append((L["kernel"], L["img_bwd"], True, self.prec))

    WEIGHT_ORDER = ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_var", "alpha")
    # Order of one layer's arrays in get_weights() / set_weights() (model.get_weights() at gcn.py:383 feeds the .npz dump;
    # PARITY UNPINNED: neither Keras nor Spektral is importable here, these restate their published code):
    #  "layer"  every layer as Dense -> BatchNormalization -> PReLU would list it: kernel, bias, gamma, beta, moving_mean,
    #           moving_variance, alpha.  That IS the Keras order for the pre / post MLPs (a Sequential of separate
    #           Dense / BN / Dropout / PReLU layers, each contributing layer.weights in turn).
    #  "keras"  (default) the same, except that GeneralConv is ONE Keras Layer with child layers: Layer.weights =
    #           trainable_weights + non_trainable_weights, and trainable_weights = the layer's own add_weight variables
    #           (kernel, bias) followed by its children's in attribute order (PReLU is created in __init__, Dropout and
    #           BatchNormalization in build): kernel, bias, alpha, gamma, beta | moving_mean, moving_variance.
    #  "keras_children_first"  the reading the round-1 review proposed (children's trainables before the layer's own):
    #           alpha, gamma, beta, kernel, bias | moving_mean, moving_variance.
    GNN_ORDERS = {"layer": WEIGHT_ORDER,
                  "keras": ("kernel", "bias", "alpha", "gamma", "beta", "moving_mean", "moving_var"),
                  "keras_children_first": ("alpha", "gamma", "beta", "kernel", "bias", "moving_mean", "moving_var")}

    def _weight_keys(self, L, order):
        keys = self.GNN_ORDERS[order] if L["group"] == "gnn" else self.WEIGHT_ORDER
        return [k for k in keys if k in L]

    def get_weights(self, order="keras"):
        return [L[k].numpy() for L in self.layers for k in self._weight_keys(L, order)]

    def set_weights(self, weights, order="keras"):
        it = iter(weights)
        for L in self.layers:
            for k in self._weight_keys(L, order):
                L[k].copy_from_host(np.asarray(next(it), np.float32).reshape(L[k].shape))

    @property
    def trainable_variables(self):
        return [L[k] for L in self.layers for k in ("kernel", "bias", "gamma", "beta", "alpha") if k in L]

    @property
    def losses(self):
        return []

    # ---- buffers ---------------------------------------------------------------------------------
    def _ensure(self, batch):
        if not self.built:
            self.build(batch.f)
        key = (batch.n, batch.n_graphs)
        if self._bufs is not None and self._bufs["key"] == key:
            return self._bufs
        ctx, n, b, h = self.ctx, batch.n, batch.n_graphs, self.hidden
        self._drop_graphs()
        wcat = h * (self.mp + 1)
        if getattr(self, "_cap", None) is None:
            self._cap = _Capacity(ctx)
        v = self._cap.view
        wp = self._wpool
        bufs = {"key": key, "cat": v("cat", n, wcat), "dcat": v("dcat", n, wcat), "h": v("h", n, h), "dh": v("dh", n, h),
                "pooled": v("pooled", b, wp), "dpooled": v("dpooled", b, wp), "probs": v("probs", b, self.output),
                "dlogits": v("dlogits", b, self.output), "zy": v("zy", b, self.output, zero=True)}
        for i, L in enumerate(self.layers):
            rows = b if L["group"] == "post" else n
            bufs[f"z{i}"] = v(f"z{i}", rows, L["fo"])             # Dense output (pre-BN), kept for the backward pass
            bufs[f"y{i}"] = v(f"y{i}", rows, L["fo"])             # layer output where it is not a slice of `cat`
        self._bufs = bufs
        return bufs

    def _multi(self):
        return self.comm is not None and self.comm.world_size > 1

    def _drop(self, L, t):
        """The layer's Dropout factor applied to t in place (forward: the activation's output -- act(s u) = s act(u) for the
        activations built; backward: the incoming gradient).  Stream = the layer's index, step = the device step count."""
        seed = self.seed + (0x9E37 * self.comm.rank if self._multi() else 0)
        D.dropout(self.ctx, t, self.dropout, seed, L["index"], self._step_dev)

    def _dense_bn(self, L, x, z, y, training):
        self._dense_bn_core(L, x, z, y, training)
        if training and self.dropout > 0.0:
            self._drop(L, y)

    def _dense_bn_core(self, L, x, z, y, training):
        ctx = self.ctx
        if not self.batch_norm:
            if not ("img_fwd" in L and self._images_fresh and
                    D.gemm_wimage(ctx, x, L["img_fwd"], L["fi"], L["fo"], z, bias=L["bias"], prec=self.prec) is not None):
                D.gemm(ctx, x, L["kernel"], L["bias"], z, prec=self._prec_of(L))
            D.bn_act(ctx, z, L["mean"], L["inv"], L["bn_gamma"], L["bn_beta"], y, act=L["act"], alpha=L.get("alpha"))
            return
        if "img_fwd" in L and self._images_fresh:
            # split-bf16 panel GEMM (csrc/gemm_panel.hip); in single-device training its epilogue also leaves the batch-norm
            # statistics of what it writes, as (rows, mean, M2) per workgroup: no pass over z for the moments
            fused_bn = training and not self._multi() and L["fo"] <= 256
            parts = self._bn_parts(x.shape[0], L["fo"]) if fused_bn else None
            nparts = D.gemm_wimage(ctx, x, L["img_fwd"], L["fi"], L["fo"], z, bias=L["bias"], prec=self.prec, bn_parts=parts)
            if nparts is not None:
                if fused_bn:
                    D.bn_finalize_parts(ctx, parts, nparts, L["mean"], L["inv"], L["moving_mean"], L["moving_var"])
                    D.bn_act(ctx, z, L["mean"], L["inv"], L["gamma"], L["beta"], y, act=L["act"], alpha=L.get("alpha"))
                    return
            else:
                D.gemm(ctx, x, L["kernel"], L["bias"], z, prec=self._prec_of(L))
        else:
            D.gemm(ctx, x, L["kernel"], L["bias"], z, prec=self._prec_of(L))
        if training and self._multi():
            # sync-BN: the two moment passes with their column sums all-reduced and the GLOBAL row count
            count = self._counts["b" if L["group"] == "post" else "n"]
            D.bn_stats(ctx, z, L["sums"])
            self.comm.allreduce_sum(L["sums"])
            D.bn_finalize(ctx, L["sums"], count, L["mean"], L["inv"])
            D.bn_stats(ctx, z, L["sums"], shift=L["mean"])
            self.comm.allreduce_sum(L["sums"])
            D.bn_finalize(ctx, L["sums"], count, L["mean"], L["inv"], L["moving_mean"], L["moving_var"], shift=L["mean"])
        elif training:
            D.bn_moments(ctx, z, L["sums"], L["mean"], L["inv"], L["moving_mean"], L["moving_var"])
        else:
            D.bn_finalize(ctx, None, 1, L["mean"], L["inv"], L["moving_mean"], L["moving_var"])
        D.bn_act(ctx, z, L["mean"], L["inv"], L["gamma"], L["beta"], y, act=L["act"], alpha=L.get("alpha"))

    def _agg_operator(self, batch):
        """The aggregation's operator: the 0 / 1 pattern of a (aggregate = "sum"), or the same pattern with 1 / (entries of
        the row) on every entry ("mean": tf.math.unsorted_segment_mean over a row's messages), built once per batch."""
        return batch.a.row_mean() if self.aggregate == "mean" else batch.a.unweighted()     # ("max" / "min": the pattern only)

    def _forward(self, batch, bufs, training):
        h, mp = self.hidden, self.mp
        cat = bufs["cat"]
        a = self._agg_operator(batch)                     # GeneralConv ignores adjacency values (8.A.4)
        x = batch.x
        li = 0
        for k in range(self.n_pre):
            L = self.layers[li]
            y = cat.cols(mp * h, (mp + 1) * h) if k == self.n_pre - 1 else bufs[f"y{li}"]
            self._dense_bn(L, x, bufs[f"z{li}"], y, training)
            x = y; li += 1
        sumc = self.connectivity == "sum"          # "sum": slice (mp - k) of `cat` holds out_k = z_k + out_(k-1) instead of z_k
        for k in range(mp):
            L = self.layers[li]
            inp = cat.cols((mp - k) * h, (mp - k + 1) * h if sumc else (mp + 1) * h)
            minmax = self.aggregate in ("max", "min", "prod")
            # ("max" / "min": the layer's messages, its aggregate and the tie counts are kept for the gradient in per-layer
            # buffers -- `h` is reused by the next layer, the `cat` slice summed over with connectivity "sum")
            hk = self._tmp(bufs, f"msg{li}", (batch.n, h)) if minmax else bufs["h"]
            self._dense_bn(L, inp, bufs[f"z{li}"], hk, training)
            new = cat.cols((mp - k - 1) * h, (mp - k) * h)
            if minmax:                                    # tf.math.unsorted_segment_max / _min over a row's messages
                ck = self._tmp(bufs, f"aggcnt{li}", (batch.n, h))
                zk = self._tmp(bufs, f"agg{li}", (batch.n, h)) if sumc else new
                D.spmm_minmax(self.ctx, a, hk, zk, ck, self.aggregate)
                if sumc:
                    D.add(self.ctx, zk, inp, new)
            else:
                D.spmm(self.ctx, a, bufs["h"], None, new)
                if sumc:
                    D.add(self.ctx, new, inp, new)
            li += 1
        if self.pool == "max":
            if getattr(self, "_cap", None) is None:
                self._cap = _Capacity(self.ctx)
            bufs["pool_arg"] = self._cap.view("pool_arg", bufs["pooled"].shape[0], bufs["pooled"].shape[1], np.int32)
        D.segment_pool(self.ctx, batch.seg, cat.cols(0, h) if sumc else cat, bufs["pooled"], self.pool,
                       bufs.get("pool_arg") if self.pool == "max" else None)
        x = bufs["pooled"]
        for k in range(self.n_post):
            L = self.layers[li]
            self._dense_bn(L, x, bufs[f"z{li}"], bufs[f"y{li}"], training)
            x = bufs[f"y{li}"]; li += 1
        return x                                          # [B, output]: BN output = the softmax logits

    def _bwd_dense_bn(self, L, dy, x, z, dx, training, accumulate=False):
        ctx = self.ctx
        dz = dy                                            # in place
        if training and self.dropout > 0.0:
            self._drop(L, dy)                              # Dropout backward: the forward's factor on the incoming gradient
        bn = self.batch_norm
        if not bn:
            D.bn_act_bwd(ctx, dy, z, L["mean"], L["inv"], L["bn_gamma"], L["bn_beta"], dz, L["scratch"], act=L["act"],
                         alpha=L.get("alpha"), training=False, dgamma=L["bn_g_gamma"], dbeta=L["bn_g_beta"],
                         dalpha=L.get("g_alpha"))
        elif training and self._multi():
            # local column sums -> parameter gradients (summed over ranks by the final all-reduce with the rest);
            # the same sums, all-reduced, and the global row count give dz
            count = self._counts["b" if L["group"] == "post" else "n"]
            D.bn_act_bwd_stats(ctx, dy, z, L["mean"], L["inv"], L["gamma"], L["beta"], L["scratch"], act=L["act"],
                               alpha=L.get("alpha"), dgamma=L["g_gamma"], dbeta=L["g_beta"], dalpha=L.get("g_alpha"))
            self.comm.allreduce_sum(L["scratch"])
            D.bn_act_bwd_apply(ctx, dy, z, L["mean"], L["inv"], L["gamma"], L["beta"], L["scratch"], count, dz,
                               act=L["act"], alpha=L.get("alpha"), training=True)
        else:
            D.bn_act_bwd(ctx, dy, z, L["mean"], L["inv"], L["gamma"], L["beta"], dz, L["scratch"], act=L["act"],
                         alpha=L.get("alpha"), training=training, dgamma=L["g_gamma"], dbeta=L["g_beta"],
                         dalpha=L.get("g_alpha"))
        if training and bn:
            # The Dense bias under a training-mode BatchNorm has the gradient sum_rows dz = gamma inv (sum g - n mean(g) -
            # mean(g xhat) sum xhat) = 0 identically (sum xhat = 0; with sync-BN the sums are the global ones): TensorFlow
            # evaluates that sum and returns rounding noise of order 1e-8; every training path here writes the exact value
            # instead of passing over dz once more (14 column-sum launches per step in the fp32 configuration).
            pass                                           # (loss_and_grads zeroes the whole flat gradient buffer once per step)
        else:
            D.act_bias_grad(ctx, dz, None, dz, None, db=L["g_bias"])
        if "img_fwd" in L and self._images_fresh and (dx is None or "img_bwd" in L):
            D.gemm_dw(ctx, x, dz, L["g_kernel"], prec=self.prec)           # (fi = 256 p: panels of the streaming kernel)
            if dx is None:
                return
            if D.gemm_wimage(ctx, dz, L["img_bwd"], L["fi"], L["fo"], dx, transpose=True, prec=self.prec, accumulate=accumulate) is not None:
                return
            D.gemm_dx(ctx, dz, L["kernel"], dx, prec=self.prec, accumulate=accumulate)
            return
        prec = self._prec_of(L)
        if dx is not None and not accumulate:
            # both products of the layer in one launch pair (gcnx_dense_bwd; falls back inside for ragged widths)
            D.dense_bwd(ctx, x, dz, L["kernel"], dx, L["g_kernel"], prec=prec)
        else:
            D.gemm_dw(ctx, x, dz, L["g_kernel"], prec=prec)
            if dx is not None:
                D.gemm_dx(ctx, dz, L["kernel"], dx, prec=prec, accumulate=accumulate)

    def _backward(self, batch, bufs, training=True):
        h, mp = self.hidden, self.mp
        cat, dcat = bufs["cat"], bufs["dcat"]
        at = self._agg_operator(batch).transpose()
        n_layers = len(self.layers)
        li = n_layers - 1
        d = bufs["dlogits"]                               # d(BN output of the last layer) from softmax+CCE
        for k in reversed(range(self.n_post)):
            L = self.layers[li]
            x = bufs["pooled"] if k == 0 else bufs[f"y{li - 1}"]
            dx = bufs["dpooled"] if k == 0 else bufs[f"y{li - 1}"]     # y of the previous layer is dead: reuse as its dY
            if k > 0:
                dx = self._tmp(bufs, f"dy{li - 1}", x.shape)
            self._bwd_dense_bn(L, d, x, bufs[f"z{li}"], dx, training)
            d = dx; li -= 1
        sumc = self.connectivity == "sum"
        D.segment_pool_bwd(self.ctx, batch.seg, bufs["dpooled"], dcat.cols(0, h) if sumc else dcat, self.pool,
                           bufs.get("pool_arg") if self.pool == "max" else None)
        for k in reversed(range(mp)):
            L = self.layers[li]
            dout = dcat.cols((mp - k - 1) * h, (mp - k) * h)
            if self.aggregate in ("max", "min", "prod"):
                zk = bufs[f"agg{li}"] if sumc else cat.cols((mp - k - 1) * h, (mp - k) * h)
                D.spmm_minmax_bwd(self.ctx, at, bufs[f"msg{li}"], zk, bufs[f"aggcnt{li}"], dout, bufs["dh"], self.aggregate)
            else:
                D.spmm(self.ctx, at, dout, None, bufs["dh"])
            if sumc:                                       # d out_(k-1) = d out_k (the skip) + dz W^T
                inp, din = cat.cols((mp - k) * h, (mp - k + 1) * h), dcat.cols((mp - k) * h, (mp - k + 1) * h)
                self._bwd_dense_bn(L, bufs["dh"], inp, bufs[f"z{li}"], din, training)
                D.add(self.ctx, din, dout, din)
            else:
                inp = cat.cols((mp - k) * h, (mp + 1) * h)
                self._bwd_dense_bn(L, bufs["dh"], inp, bufs[f"z{li}"], dcat.cols((mp - k) * h, (mp + 1) * h), training,
                                   accumulate=True)
            li -= 1
        d = dcat.cols(mp * h, (mp + 1) * h)
        for k in reversed(range(self.n_pre)):
            L = self.layers[li]
            x = batch.x if k == 0 else bufs[f"y{li - 1}"]
            dx = None if k == 0 else self._tmp(bufs, f"dy{li - 1}", x.shape)
            self._bwd_dense_bn(L, d, x, bufs[f"z{li}"], dx, training)
            d = dx; li -= 1

    def _prec_of(self, L):
        """Arithmetic of a layer's products outside the panel kernels: the post-MLP runs on B rows (32 graphs): a few MFLOP
        for which the bf16 tile kernels' operand images cost more than the products (28 us per launch against the fp32
        tiles' few) -- exact fp32 there."""
        return "f32" if L["group"] == "post" else self.prec

    def _bn_parts(self, rows, fo):
        """Scratch for one layer's batch-norm parts [(rows, mean, M2) x workgroups x fo] (consumed before the next layer runs)."""
        need = 3 * max(D.gemm_wimage_parts(self.ctx, rows), 1) * fo
        cur = getattr(self, "_bn_parts_buf", None)
        if cur is None or cur.size < need:
            self._bn_parts_buf = self.ctx.empty(need)
        return self._bn_parts_buf

    def _prepare_images(self):
        """The weight images of this step's forward and backward products, all in one launch."""
        if getattr(self, "_img_prec", None) != self.prec:       # (the precision was switched after build)
            self._alloc_images()
        self._images_fresh = bool(self._img_jobs)
        if self._img_jobs:
            D.wimage_prepare(self.ctx, self._img_jobs)

    def _tmp(self, bufs, key, shape):
        if key not in bufs or bufs[key].shape != tuple(shape):
            if getattr(self, "_cap", None) is None:
                self._cap = _Capacity(self.ctx)
            shape = tuple(shape) if len(shape) == 2 else (1, int(np.prod(shape)))
            bufs[key] = self._cap.view("tmp_" + key, shape[0], shape[1])
            if len(shape) != 2:
                bufs[key] = bufs[key].flat(0, shape[1])
        return bufs[key]

    # ---- public surface ----------------------------------------------------------------------------
    def _as_batch(self, inputs, target=None):
        if isinstance(inputs, DeviceBatch):
            if target is not None and inputs.y is None:
                inputs.y = self.ctx.to_device(target, np.float32)
            return inputs
        return DeviceBatch.from_host(self.ctx, inputs, target, weighted=False)

    def __call__(self, inputs, training=False):
        batch = self._as_batch(inputs)
        bufs = self._ensure(batch)
        self._prepare_images()
        logits = self._forward(batch, bufs, training)
        if self.activation is None:                        # linear head: the last BatchNormalization's output
            return logits.numpy()
        la = self._tmp(bufs, "la_scratch", (2,))
        la.fill_zero()
        D.softmax_cce(self.ctx, logits, bufs["zy"], bufs["probs"], la, None, None)
        return bufs["probs"].numpy()

    def loss_and_grads(self, inputs, target=None, _lr=None, global_batch=None):
        """Forward (training=True) + loss + every gradient.  With a communicator the batch is this rank's shard:
        the loss is normalised by ``global_batch`` graphs, BatchNorm runs on the global statistics, and one
        all-reduce sums the flat gradient buffer (+ loss / accuracy tail) over the ranks."""
        batch = self._as_batch(inputs, target)
        bufs = self._ensure(batch)
        multi = self._multi()
        denom = float(global_batch or batch.n_graphs)
        if multi:
            # global row / graph counts for sync-BN: one host round trip per BATCH (cached by its uid), not per step
            if getattr(self, "_counts_uid", None) != batch.uid:
                tot = self.comm.allreduce_host([batch.n, batch.n_graphs], "sum")
                self._counts = {"n": float(tot[0]), "b": float(tot[1])}
                self._counts_uid = batch.uid
            denom = float(global_batch or self._counts["b"])
        # with a capturable communicator (RCCL) the whole sync-BN step -- 2 small all-reduces per layer forward, 1 per
        # layer backward, the gradient all-reduce and SGD -- is recorded into ONE HIP graph
        fused_comm = (multi and _lr is not None and self.use_graph and getattr(self.comm, "capturable", False)
                      and not getattr(self, "_comm_capture_failed", False))

        def seq():
            self._prepare_images()
            logits = self._forward(batch, bufs, True)
            self.flat_g.fill_zero()                        # loss / accuracy tail and the analytically zero bias gradients: one memset
            D.softmax_cce(self.ctx, logits, batch.y, bufs["probs"], self.loss_acc, bufs["dlogits"], denom, cce=self.cce_train)
            self._backward(batch, bufs, True)
            if fused_comm:
                self.comm.allreduce_sum(self.flat_g)
            if _lr is not None and (fused_comm or not multi):   # the update rides in the same captured graph
                D.sgd(self.ctx, self.flat_p, self.flat_g.flat(0, self.n_params), _lr)
            if self.dropout > 0.0:
                D.counter_add(self.ctx, self._step_dev, 1)     # the next step draws new Dropout masks (also from a replayed graph)
        self._bind(batch)
        self._step_applied = fused_comm or not multi
        lr_key = self._lr_key(_lr)
        if multi and not fused_comm:
            seq()                                          # host-mediated collectives inside: not captured
            self.comm.allreduce_sum(self.flat_g)
        else:
            try:
                self._run(("grad", batch.uid, lr_key, denom), seq)
            except Exception as e:
                if not fused_comm:
                    raise
                import sys
                print(f"gcnx: the sync-BN step could not be captured with its collectives ({e}); running it eagerly",
                      file=sys.stderr)
                self._comm_capture_failed = True
                self._drop_graphs()
                return self.loss_and_grads(batch, None, _lr, global_batch)
        self.ctx.set_lr_source(None)
        return batch

    def train_step(self, inputs, target=None, lr=0.02, fetch=True, global_batch=None):
        """gcn.py:330-340 for the live model: forward(training=True), CCE, gradients, SGD, accuracy."""
        batch = self.loss_and_grads(inputs, target, _lr=float(lr), global_batch=global_batch)
        if not self._step_applied:
            D.sgd(self.ctx, self.flat_p, self.flat_g.flat(0, self.n_params), lr)
        n_graphs = global_batch or (self._counts["b"] if self._multi() else batch.n_graphs)
        if fetch == "stash":
            self.stash_metrics(n_graphs)
            return None
        if not fetch:
            return None
        la = self.loss_acc.numpy()
        return float(la[0]), float(la[1]) / float(n_graphs)

    def evaluate_batch(self, inputs, target):
        batch = self._as_batch(inputs, target)
        bufs = self._ensure(batch)
        self._prepare_images()
        logits = self._forward(batch, bufs, False)
        self.loss_acc.fill_zero()
        D.softmax_cce(self.ctx, logits, batch.y, bufs["probs"], self.loss_acc, None, batch.n_graphs, cce=self.cce_eval)
        la = self.loss_acc.numpy()
        return float(la[0]), float(la[1]) / batch.n_graphs, (logits if self.activation is None else bufs["probs"]).numpy()

    def gradients(self):
        return [{k[2:]: L[k].numpy() for k in L if k.startswith("g_")} for L in self.layers]
