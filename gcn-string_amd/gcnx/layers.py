"""Host mirror of the Spektral / Keras layer call surface used on the hot path.

Upstream signatures kept: ``GCNConv(channels, activation=None, use_bias=True,
kernel_initializer="glorot_uniform", bias_initializer="zeros")`` called as ``layer([x, a])``;
``GCNConv.preprocess(a)``; ``GlobalSumPool()([x, i])`` (+ Avg/Max); ``Dense(units, activation)``.
Reference topology: GCNConv -> GCNConv -> global pool -> Linear (gcn_utills.py:805-808,
832-842); live model ctor gcn.py:320, forward gcn.py:334/351, gradients gcn.py:337.

There is no autograd here: every layer has ``backward(dy)`` that returns dx and leaves the
parameter gradients in ``layer.grads`` (what tape.gradient, gcn.py:337, would produce).
All arithmetic runs in libgcnx (HIP); these classes only own buffers and sequence calls.
"""
from __future__ import annotations

import os

import numpy as np

from . import device as D

_FUSED_KNOB = os.environ.get("GCNX_FUSED", "1") != "0"     # tuning knob, read once at import (diagnostics)


def glorot_uniform(rng, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


class Layer:
    """Keras-like lazy build: parameters are created on the first call, when the input width
    is known.  ``storage`` lets a model place all parameters in one flat buffer."""

    def __init__(self, ctx=None, seed=None):
        self.ctx = ctx
        self.built = False
        self.params, self.grads = {}, {}
        self._rng = np.random.default_rng(seed)
        self._seed = 0 if seed is None else int(seed)    # of the layer's Dropout streams
        self._scratch = {}

    # parameter spec: list of (name, shape, initial host array)
    def _param_spec(self, in_dim):
        return []

    def n_params(self, in_dim):
        return sum(int(np.prod(s)) for _, s, _ in self._param_spec(in_dim))

    def build(self, ctx, in_dim, p_store=None, g_store=None, offset=0):
        self.ctx = ctx
        spec = self._param_spec(in_dim)
        total = sum(int(np.prod(s)) for _, s, _ in spec)
        if p_store is None:
            p_store, g_store, offset = ctx.zeros(max(total, 1)), ctx.zeros(max(total, 1)), 0
        off = offset
        for name, shape, init in spec:
            n = int(np.prod(shape))
            self.params[name] = p_store.flat(off, n, shape)
            self.grads[name] = g_store.flat(off, n, shape)
            self.params[name].copy_from_host(init)
            off += n
        self.in_dim, self.built = in_dim, True
        return off

    def _buf(self, key, shape, dtype=np.float32):
        b = self._scratch.get(key)
        if b is None or b.shape != tuple(shape):
            b = self.ctx.empty(shape, dtype)
            self._scratch[key] = b
        return b

    def get_weights(self):
        return [self.params[k].numpy() for k in self.params]

    def set_weights(self, weights):
        for k, w in zip(self.params, weights):
            self.params[k].copy_from_host(w)

    @property
    def trainable_variables(self):
        return list(self.params.values())

    def __call__(self, inputs, **kw):
        return self.call(inputs, **kw)


class GCNConv(Layer):
    """out = activation(A^ (x W) + b)   -- bias after aggregation."""

    def __init__(self, channels, activation=None, use_bias=True, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", prec="f32", **kw):
        super().__init__(**kw)
        if activation not in (None, "linear", "relu"):
            raise NotImplementedError(f"GCNConv activation {activation!r}: only None/'relu' are fused in the SpMM epilogue")
        if kernel_initializer != "glorot_uniform" or bias_initializer != "zeros":
            raise NotImplementedError("only glorot_uniform / zeros initialisers (the Spektral defaults)")
        self.channels, self.activation, self.use_bias, self.prec = int(channels), activation, use_bias, prec

    @staticmethod
    def preprocess(a, mode="spektral"):
        """gcn_filter on one graph's scipy adjacency (a dataset transform in Spektral): adds I
        unconditionally (diagonal 2 where a self-loop exists), D^-1/2 A~ D^-1/2 (8.A.2)."""
        import scipy.sparse as sp

        a = sp.csr_matrix(a, dtype=np.float64)
        if mode == "spektral":
            a = a + sp.identity(a.shape[0], format="csr")
        else:  # PyG add_remaining_self_loops
            a = a + sp.diags(np.where(a.diagonal() == 0, 1.0, 0.0))
        a = sp.csr_matrix(a)
        deg = np.asarray(a.sum(1)).ravel()
        dinv = np.zeros_like(deg)
        dinv[deg > 0] = 1.0 / np.sqrt(deg[deg > 0])
        out = sp.csr_matrix(sp.diags(dinv) @ a @ sp.diags(dinv))
        out.sort_indices()
        return out

    def _param_spec(self, in_dim):
        spec = [("kernel", (in_dim, self.channels), glorot_uniform(self._rng, in_dim, self.channels))]
        if self.use_bias:
            spec.append(("bias", (self.channels,), np.zeros(self.channels, np.float32)))
        return spec

    def call(self, inputs, out=None):
        x, a = inputs
        if not self.built:
            self.build(x.ctx, x.shape[1])
        n = x.shape[0]
        y = out if out is not None else self._buf("y", (n, self.channels))
        if self._one_launch(x, a):
            # small-feature regime: (A x) W in one launch (csrc/fused.hip); S = A x is kept for dW = S^T dZ
            s = self._buf("s", (n, x.shape[1]))
            D.gcn_conv_fwd(self.ctx, a, x, self.params["kernel"], self.params.get("bias"), y, act=self.activation, s=s,
                           prec=self.prec)
            self._saved = (x, a, y, s)
            return y
        h = self._buf("h", (n, self.channels))
        D.gemm(self.ctx, x, self.params["kernel"], None, h, prec=self.prec)
        D.spmm(self.ctx, a, h, self.params.get("bias"), y, act=self.activation)
        self._saved = (x, a, y, None)
        return y

    def _one_launch(self, x, a):
        return (self.prec in ("f32", "bf16x3") and getattr(a, "plan", None) is None and x.contiguous and _FUSED_KNOB
                and D.gcn_conv_fused_ok(self.ctx, x.shape[0], x.shape[1], self.channels, x.ld))

    def backward(self, dy, need_dx=True, dy_is_dz=False):
        """dy: gradient wrt the layer output.  dy_is_dz=True when the caller already applied the
        activation mask and filled grads['bias'] (fused upstream)."""
        x, a, y, s = self._saved
        n = x.shape[0]
        dz = dy
        if not dy_is_dz:
            dz = self._buf("dz", (n, self.channels))
            D.act_bias_grad(self.ctx, dy, y, dz, self.activation, db=self.grads.get("bias"))
        if s is not None:                        # forward was (A x) W: dW = S^T dZ, dx = A^T (dZ W^T)
            D.gemm_dw(self.ctx, s, dz, self.grads["kernel"], prec=self.prec)
            if not need_dx:
                return None
            t = self._buf("t", (n, self.in_dim))
            D.gemm_dx(self.ctx, dz, self.params["kernel"], t, prec=self.prec)
            dx = self._buf("dx", (n, self.in_dim))
            D.spmm(self.ctx, a.transpose(), t, None, dx)
            return dx
        dh = self._buf("h", (n, self.channels))  # forward scratch is dead by now
        D.spmm(self.ctx, a.transpose(), dz, None, dh)
        D.gemm_dw(self.ctx, x, dh, self.grads["kernel"], prec=self.prec)
        if not need_dx:
            return None
        dx = self._buf("dx", (n, self.in_dim))
        D.gemm_dx(self.ctx, dh, self.params["kernel"], dx, prec=self.prec)
        return dx


class GeneralConv(Layer):
    """spektral.layers.GeneralConv -- the message-passing layer inside the model the reference trains (gcn.py:320
    -> GeneralGNN):

        h = activation(BatchNormalization(x W + b));   out[t] = sum_{(t, s) in a.indices} h[s]

    ``GeneralConv(channels=256, batch_norm=True, dropout=0.0, aggregate="sum", activation="prelu", use_bias=True)``,
    called as ``layer([x, a], training=bool)``.  The adjacency VALUES are ignored, no self-loop is added and nothing
    is normalised (Spektral's ``propagate`` gathers by a.indices and segment-sums).  Only what the reference uses is
    built: aggregate="sum", dropout=0.0; activation "prelu" (per-feature slopes, initial 0), "relu" or None.
    Keras BatchNormalization semantics: momentum 0.99, eps 1e-3, biased batch variance, moving statistics updated in
    training.  ``backward(dy)`` returns dx and leaves kernel / bias / gamma / beta / alpha gradients in ``grads``.
    get_weights() order: kernel, bias, alpha, gamma, beta, moving_mean, moving_variance (the layer's own variables,
    then its children's; non-trainables last -- see GeneralGNN.GNN_ORDERS; PARITY UNPINNED)."""

    def __init__(self, channels=256, batch_norm=True, dropout=0.0, aggregate="sum", activation="prelu", use_bias=True,
                 prec="f32", **kw):
        super().__init__(**kw)
        if aggregate not in ("sum", "mean", "max", "min", "prod"):
            raise ValueError(f"GeneralConv(aggregate={aggregate!r}): Spektral's aggregations are 'sum' (what gcn.py:320 uses), 'mean', 'max', 'min', 'prod'")
        self.aggregate = aggregate
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"GeneralConv(dropout={dropout!r}): a rate in [0, 1)")
        if activation not in (None, "linear", "relu", "prelu"):
            raise NotImplementedError(f"GeneralConv activation {activation!r}")
        self.channels, self.batch_norm, self.activation, self.use_bias, self.prec = int(channels), bool(batch_norm), activation, use_bias, prec
        # Dropout(rate) between BatchNormalization and the activation (8.A.4), training only: applied as its factor on the
        # activation's output (act(s u) = s act(u) for PReLU / ReLU / linear) and on the incoming gradient; the mask of call
        # number t is a stateless hash of (seed, t) -- this library's generator, not TensorFlow's
        self.dropout, self._drop_calls = float(dropout), 0
        self.state = {}

    def _param_spec(self, in_dim):
        c = self.channels
        spec = [("kernel", (in_dim, c), glorot_uniform(self._rng, in_dim, c))]
        if self.use_bias:
            spec.append(("bias", (c,), np.zeros(c, np.float32)))
        if self.activation == "prelu":
            spec.append(("alpha", (c,), np.zeros(c, np.float32)))
        if self.batch_norm:
            spec += [("gamma", (c,), np.ones(c, np.float32)), ("beta", (c,), np.zeros(c, np.float32))]
        return spec

    def build(self, ctx, in_dim, p_store=None, g_store=None, offset=0):
        off = super().build(ctx, in_dim, p_store, g_store, offset)
        if self.batch_norm:
            c = self.channels
            self.state = {"moving_mean": ctx.zeros(c), "moving_var": ctx.to_device(np.ones(c, np.float32))}
            self._mean, self._inv, self._sums, self._bn_scratch = ctx.zeros(c), ctx.zeros(c), ctx.zeros(2 * c), ctx.zeros(3 * c)
        elif self.activation == "prelu":
            # PReLU without BatchNormalization: the fused batch-norm + activation pass with the identity transform
            c = self.channels
            self._mean, self._inv, self._bn_scratch = ctx.zeros(c), ctx.to_device(np.ones(c, np.float32)), ctx.zeros(3 * c)
            self._dummy = (ctx.zeros(c), ctx.zeros(c))
        return off

    def get_weights(self):
        return [self.params[k].numpy() for k in self.params] + [self.state[k].numpy() for k in self.state]

    def set_weights(self, weights):
        for k, w in zip(list(self.params) + list(self.state), weights):
            (self.params[k] if k in self.params else self.state[k]).copy_from_host(np.asarray(w, np.float32))

    def call(self, inputs, training=False, out=None):
        x, a = inputs
        if not self.built:
            self.build(x.ctx, x.shape[1])
        ctx, n, c = self.ctx, x.shape[0], self.channels
        z, h = self._buf("z", (n, c)), self._buf("h", (n, c))
        ident = not self.batch_norm and self.activation == "prelu"
        D.gemm(ctx, x, self.params["kernel"], self.params.get("bias"), z, prec=self.prec,
               act=None if (self.batch_norm or ident) else self.activation)
        if ident:
            D.bn_act(ctx, z, self._mean, self._inv, self._inv, self._mean, h, act="prelu", alpha=self.params["alpha"])
        elif self.batch_norm:
            if training:
                D.bn_moments(ctx, z, self._sums, self._mean, self._inv, self.state["moving_mean"], self.state["moving_var"])
            else:
                D.bn_finalize(ctx, None, 1, self._mean, self._inv, self.state["moving_mean"], self.state["moving_var"])
            D.bn_act(ctx, z, self._mean, self._inv, self.params["gamma"], self.params["beta"], h, act=self.activation,
                     alpha=self.params.get("alpha"))
        else:
            h = z
        drop_id = None
        if training and self.dropout > 0.0:
            drop_id = self._drop_calls = self._drop_calls + 1
            D.dropout(ctx, h, self.dropout, self._seed, drop_id)
        y = out if out is not None else self._buf("y", (n, c))
        au = a.row_mean() if self.aggregate == "mean" else a.unweighted()  # values ignored (8.A.4); "mean": 1 / row length
        cnt = None
        if self.aggregate in ("max", "min", "prod"):        # unsorted_segment_max / _min / _prod; + what their gradients need
            cnt = self._buf("aggcnt", (n, c))
            D.spmm_minmax(ctx, au, h, y, cnt, self.aggregate)
        else:
            D.spmm(ctx, au, h, None, y)
        self._saved = (x, au, z, h, bool(training), drop_id, y, cnt)
        return y

    def backward(self, dy, need_dx=True):
        x, au, z, h, training, drop_id, y, cnt = self._saved
        ctx = self.ctx
        dh = self._buf("dh", dy.shape)
        if cnt is not None:
            D.spmm_minmax_bwd(ctx, au.transpose(), h, y, cnt, dy, dh, self.aggregate)
        else:
            D.spmm(ctx, au.transpose(), dy, None, dh)        # dH = S^T dY
        if drop_id is not None:
            D.dropout(ctx, dh, self.dropout, self._seed, drop_id)
        if not self.batch_norm and self.activation == "prelu":
            D.bn_act_bwd(ctx, dh, z, self._mean, self._inv, self._inv, self._mean, dh, self._bn_scratch, act="prelu",
                         alpha=self.params["alpha"], training=False, dgamma=self._dummy[0], dbeta=self._dummy[1],
                         dalpha=self.grads["alpha"])
            D.act_bias_grad(ctx, dh, None, dh, None, db=self.grads.get("bias"))
        elif self.batch_norm:
            D.bn_act_bwd(ctx, dh, z, self._mean, self._inv, self.params["gamma"], self.params["beta"], dh, self._bn_scratch,
                         act=self.activation, alpha=self.params.get("alpha"), training=training,
                         dgamma=self.grads["gamma"], dbeta=self.grads["beta"], dalpha=self.grads.get("alpha"))
            D.act_bias_grad(ctx, dh, None, dh, None, db=self.grads.get("bias"))
        else:
            D.act_bias_grad(ctx, dh, h, dh, self.activation, db=self.grads.get("bias"), alpha=self.params.get("alpha"),
                            dalpha=self.grads.get("alpha"))
        D.gemm_dw(ctx, x, dh, self.grads["kernel"], prec=self.prec)
        if not need_dx:
            return None
        dx = self._buf("dx", x.shape)
        D.gemm_dx(ctx, dh, self.params["kernel"], dx, prec=self.prec)
        return dx


class Dense(Layer):
    """Keras Dense: act(x W + b); activation None / 'relu' (softmax is fused with the loss)."""

    def __init__(self, units, activation=None, use_bias=True, prec="f32", **kw):
        super().__init__(**kw)
        if activation not in (None, "linear", "relu"):
            raise NotImplementedError(f"Dense activation {activation!r}")
        self.units, self.activation, self.use_bias, self.prec = int(units), activation, use_bias, prec

    def _param_spec(self, in_dim):
        spec = [("kernel", (in_dim, self.units), glorot_uniform(self._rng, in_dim, self.units))]
        if self.use_bias:
            spec.append(("bias", (self.units,), np.zeros(self.units, np.float32)))
        return spec

    def call(self, x, out=None):
        if not self.built:
            self.build(x.ctx, x.shape[1])
        y = out if out is not None else self._buf("y", (x.shape[0], self.units))
        D.gemm(self.ctx, x, self.params["kernel"], self.params.get("bias"), y, act=self.activation, prec=self.prec)
        self._saved = (x, y)
        return y

    def backward(self, dy, need_dx=True):
        x, y = self._saved
        dz = self._buf("dz", dy.shape)
        D.act_bias_grad(self.ctx, dy, y, dz, self.activation, db=self.grads.get("bias"))
        D.gemm_dw(self.ctx, x, dz, self.grads["kernel"], prec=self.prec)
        if not need_dx:
            return None
        dx = self._buf("dx", (x.shape[0], self.in_dim))
        D.gemm_dx(self.ctx, dz, self.params["kernel"], dx, prec=self.prec)
        return dx


class _GlobalPool(Layer):
    mode = "sum"

    def call(self, inputs, out=None):
        x, seg = inputs
        if not isinstance(seg, D.Segments):
            seg = D.Segments.from_ids(x.ctx, seg)
        self.ctx = x.ctx
        pooled = out if out is not None else self._buf("p", (seg.n_graphs, x.shape[1]))
        arg = self._buf("arg", (seg.n_graphs, x.shape[1]), np.int32) if self.mode == "max" else None
        D.segment_pool(self.ctx, seg, x, pooled, self.mode, arg)
        self._saved = (x.shape, seg, arg)
        return pooled

    def backward(self, dp, y_mask=None, db=None, out=None):
        shape, seg, arg = self._saved
        dx = out if out is not None else self._buf("dx", shape)
        D.segment_pool_bwd(self.ctx, seg, dp, dx, self.mode, arg, y=y_mask, db=db)
        return dx


class GlobalSumPool(_GlobalPool):
    """P[g] = sum_{n: i[n]=g} X[n]  (tf.math.segment_sum; GeneralGNN pool='sum', gcn.py:320)."""
    mode = "sum"


class GlobalAvgPool(_GlobalPool):
    mode = "avg"


class GlobalMaxPool(_GlobalPool):
    """global_max_pool of the reference's torch topology (gcn_utills.py:842)."""
    mode = "max"
