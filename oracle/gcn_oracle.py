"""CPU ORACLE (numpy) for the GCN forward/backward hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference (Sum02dean/GCN-STRING) holds no native code, no
tests and no golden vectors for this path; its arithmetic lives in un-vendored,
un-pinned Spektral / TensorFlow (reference call sites: src/scripts/gcn.py:8-10
imports, :316-317 DisjointLoader, :320 GeneralGNN, :326 CategoricalCrossentropy,
:328-340 train_step, :342-362 evaluate).  Neither package is importable in the
build container or on the GPU box, so this file restates the *published*
Spektral 1.x / Keras / TF 2.x semantics (SURVEY.md section 8.A is the spec of
record) and is cross-checked in tests/ against two independent implementations
that are importable here: scipy.sparse (SpMM, normalisation) and torch-CPU fp64
autograd (every gradient).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (gcn-string_amd/) never does.

Everything is dtype-generic: pass float64 arrays for golden vectors, float32 to
emulate TF's fp32 kernels.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- #
# A.1  DisjointLoader collate  (reference use: src/scripts/gcn.py:316-317,350,367)
# --------------------------------------------------------------------------- #

def disjoint_collate(graphs):
    """Spektral ``DisjointLoader.collate`` for graph-level labels.

    ``graphs``: list of (x[n,F], a, y[n_labels]) with ``a`` a scipy sparse matrix
    (the data contract of ``MyDataset`` -- src/scripts/gcn.py:153-157: x float64,
    a int64 0/1 symmetric *with self-loops*, y one-hot).

    Returns ``(x, (indices[nnz,2] int64, values[nnz], dense_shape), i[N] int64, y[B,L])``
    where indices are (row, col) in row-major order, exactly what
    ``sp_matrix_to_sp_tensor`` + ``tf.sparse.reorder`` yield.  ``sp.find`` drops
    explicit zeros, so do we.
    """
    import scipy.sparse as sp

    x = np.vstack([g[0] for g in graphs])
    a = sp.block_diag([g[1] for g in graphs]).tocoo()
    n_nodes = np.array([g[0].shape[0] for g in graphs], dtype=np.int64)
    i = np.repeat(np.arange(len(graphs), dtype=np.int64), n_nodes)
    keep = a.data != 0
    row, col, val = a.row[keep].astype(np.int64), a.col[keep].astype(np.int64), a.data[keep]
    order = np.lexsort((col, row))  # row-major, then column (tf.sparse.reorder)
    indices = np.stack([row[order], col[order]], axis=1)
    y = np.array([g[2] for g in graphs])
    return x, (indices, val[order], (int(x.shape[0]), int(x.shape[0]))), i, y


def coo_to_csr(indices, n_rows):
    """Row-major-sorted COO (row, col) -> CSR (rowptr int64[N+1], colidx int64[nnz])."""
    rows = indices[:, 0]
    counts = np.bincount(rows, minlength=n_rows)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr, indices[:, 1].astype(np.int64)


def graph_ptr_from_ids(i, n_graphs):
    """Sorted segment ids ``i`` (A.1) -> graph_ptr[B+1] (row range of each graph)."""
    counts = np.bincount(i, minlength=n_graphs)
    gp = np.zeros(n_graphs + 1, dtype=np.int64)
    np.cumsum(counts, out=gp[1:])
    return gp


# --------------------------------------------------------------------------- #
# A.2  gcn_filter  (= GCNConv.preprocess; reference topology gcn_utills.py:805-806)
# --------------------------------------------------------------------------- #

def gcn_filter_csr(rowptr, colidx, vals, mode="spektral", dtype=np.float64):
    """Normalised adjacency for a CSR matrix whose every row stores its diagonal.

    mode "spektral": A~ = A + I unconditionally (diagonal of a self-looped graph
    becomes 2) -- Spektral ``gcn_filter``/``normalized_adjacency``.
    mode "pyg": ``add_remaining_self_loops`` -- diagonal = existing value (weight
    kept), a loop of weight 1 only where none exists.  With the precondition that
    the diagonal is stored, an existing loop is simply kept.
    deg = row sums of A~;  A^ = D^-1/2 A~ D^-1/2, with deg^-1/2 := 0 where deg == 0.
    """
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    v = np.ones(len(colidx), dtype=dtype) if vals is None else np.asarray(vals, dtype=dtype).copy()
    diag = rows == colidx
    if mode == "spektral":
        v[diag] += 1.0
    elif mode != "pyg":
        raise ValueError(mode)
    deg = np.zeros(n, dtype=dtype)
    np.add.at(deg, rows, v)
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, 1.0 / np.sqrt(np.where(deg > 0, deg, 1.0)), 0.0).astype(dtype)
    return v * dinv[rows] * dinv[colidx]


def gcn_filter_scipy(a, mode="spektral"):
    """Structural form (adds missing diagonal entries) on a scipy matrix; fp64."""
    import scipy.sparse as sp

    a = sp.csr_matrix(a, dtype=np.float64)
    n = a.shape[0]
    if mode == "spektral":
        a = a + sp.identity(n, format="csr")
    else:
        d = a.diagonal()
        a = a + sp.diags(np.where(d == 0, 1.0, 0.0))
    a = sp.csr_matrix(a)
    a.sort_indices()
    deg = np.asarray(a.sum(1)).ravel()
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, np.power(np.where(deg > 0, deg, 1.0), -0.5), 0.0)
    out = sp.diags(dinv) @ a @ sp.diags(dinv)
    out = sp.csr_matrix(out)
    out.sort_indices()
    return out


# --------------------------------------------------------------------------- #
# K2/K3  neighbour aggregation  (SparseTensorDenseMatMul / gather+segment_sum)
# --------------------------------------------------------------------------- #

def spmm_csr(rowptr, colidx, vals, h):
    """out[t] = sum_e vals[e] * h[colidx[e]] over row t's entries, in storage order.

    vals None = unweighted (GeneralConv ``propagate`` with aggregate="sum",
    A.4: adjacency values ignored).  Accumulates in h.dtype, entry by entry, which
    is what TF's CPU SparseTensorDenseMatMul / UnsortedSegmentSum do.
    """
    n = len(rowptr) - 1
    out = np.zeros((n, h.shape[1]), dtype=h.dtype)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    contrib = h[colidx] if vals is None else h[colidx] * np.asarray(vals, dtype=h.dtype)[:, None]
    np.add.at(out, rows, contrib)
    return out


# bf16 feature storage (the aggregation "fp32 and bf16 both reported").  bfloat16 is the upper half
# of an IEEE float32; TensorFlow's float32 -> bfloat16 cast rounds to nearest even (tensorflow/core/lib/bfloat16).  The
# reference itself runs fp32 throughout (gcn.py:320-340): this models the device's bf16-feature variant, PARITY UNPINNED.

def bf16_bits(x):
    """float32 -> bfloat16 bit patterns (uint16), round to nearest even; NaN not handled (never produced here)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + (((u >> 16) & 1) + np.uint32(0x7fff))) >> 16).astype(np.uint16)


def bf16_from_bits(b):
    """bfloat16 bit patterns -> the float32 values they denote (exact)."""
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def spmm_csr_bf16(rowptr, colidx, vals, h_bits, bias=None, relu=False):
    """The aggregation of spmm_csr on bf16-stored features: exact products and sums (float64) of the bf16 values, bias,
    ReLU; returns the float64 result BEFORE the final rounding (the caller rounds with bf16_bits and allows one unit in
    the last place: the device accumulates in fp32, in another order)."""
    out = spmm_csr(rowptr, colidx, None if vals is None else np.asarray(vals, np.float64), bf16_from_bits(h_bits).astype(np.float64))
    if bias is not None:
        out = out + np.asarray(bias, np.float64)
    return np.maximum(out, 0) if relu else out


def spmm_csr_T(rowptr, colidx, vals, g):
    """dH = A^T dZ :  dH[colidx[e]] += vals[e] * g[row(e)]   (K2^T / K3^T)."""
    n = len(rowptr) - 1
    out = np.zeros((n, g.shape[1]), dtype=g.dtype)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    contrib = g[rows] if vals is None else g[rows] * np.asarray(vals, dtype=g.dtype)[:, None]
    np.add.at(out, colidx, contrib)
    return out


def csr_transpose(rowptr, colidx, vals):
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    order = np.lexsort((rows, colidx))
    t_rows = colidx[order]
    t_rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(t_rows, minlength=n), out=t_rowptr[1:])
    return t_rowptr, rows[order].astype(np.int64), (None if vals is None else np.asarray(vals)[order])


# --------------------------------------------------------------------------- #
# Activations
# --------------------------------------------------------------------------- #

def act_fwd(z, act, alpha=None):
    if act in (None, "none", "linear"):
        return z
    if act == "relu":
        return np.maximum(z, 0)
    if act == "prelu":  # Keras PReLU: relu(x) - alpha*relu(-x), alpha per feature (A.5)
        return np.maximum(z, 0) + alpha * np.minimum(z, 0)
    raise ValueError(act)


def act_bwd(dy, z_or_y, act, alpha=None):
    """dZ given dY.  For relu the mask may be taken from Y (y>0 <=> z>0)."""
    if act in (None, "none", "linear"):
        return dy
    if act == "relu":
        return dy * (z_or_y > 0)
    if act == "prelu":
        return dy * np.where(z_or_y > 0, 1.0, alpha).astype(dy.dtype)
    raise ValueError(act)


# --------------------------------------------------------------------------- #
# A.4  GCNConv  out = act(A^ (x W) + b)      backward: A.6
# --------------------------------------------------------------------------- #

def gcn_conv_fwd(x, csr, w, b, act="relu"):
    rowptr, colidx, vals = csr
    h = x @ w                                  # K1
    z = spmm_csr(rowptr, colidx, vals, h)      # K2
    if b is not None:
        z = z + b                              # bias AFTER aggregation (A.4)
    y = act_fwd(z, act)
    return y, {"x": x, "h": h, "y": y}


def gcn_conv_bwd(dy, cache, csr, w, act="relu", need_dx=True, csr_t=None):
    """A.6: dZ = dY*act'; db = colsum dZ; dH = A^T dZ; dW = X^T dH; dX = dH W^T."""
    rowptr, colidx, vals = csr
    dz = act_bwd(dy, cache["y"], act)
    db = dz.sum(0)
    if csr_t is not None:
        dh = spmm_csr(csr_t[0], csr_t[1], csr_t[2], dz)
    else:
        dh = spmm_csr_T(rowptr, colidx, vals, dz)
    dw = cache["x"].T @ dh
    dx = dh @ w.T if need_dx else None
    return dx, dw, db


# --------------------------------------------------------------------------- #
# K4  Global pools  (tf.math.segment_sum / mean / max over sorted ids)
# --------------------------------------------------------------------------- #

def global_pool_fwd(x, graph_ptr, mode="sum"):
    b = len(graph_ptr) - 1
    out = np.zeros((b, x.shape[1]), dtype=x.dtype)
    arg = None
    if mode == "max":
        arg = np.zeros((b, x.shape[1]), dtype=np.int64)
    for g in range(b):
        lo, hi = int(graph_ptr[g]), int(graph_ptr[g + 1])
        if hi == lo:
            continue  # empty segment -> 0 (segment_sum); never produced by DisjointLoader
        seg = x[lo:hi]
        if mode == "sum":
            out[g] = seg.sum(0)
        elif mode == "avg":
            out[g] = seg.sum(0) / (hi - lo)
        elif mode == "max":
            arg[g] = lo + seg.argmax(0)  # first maximal row wins
            out[g] = seg.max(0)
        else:
            raise ValueError(mode)
    return out, arg


def global_pool_bwd(dp, graph_ptr, n, mode="sum", arg=None):
    dx = np.zeros((n, dp.shape[1]), dtype=dp.dtype)
    for g in range(len(graph_ptr) - 1):
        lo, hi = int(graph_ptr[g]), int(graph_ptr[g + 1])
        if hi == lo:
            continue
        if mode == "sum":
            dx[lo:hi] = dp[g]
        elif mode == "avg":
            dx[lo:hi] = dp[g] / (hi - lo)
        elif mode == "max":
            dx[arg[g], np.arange(dp.shape[1])] = dp[g]
    return dx


# --------------------------------------------------------------------------- #
# A.5  softmax + CategoricalCrossentropy + accuracy   (gcn.py:326,335,339)
# --------------------------------------------------------------------------- #

def softmax(z):
    z = z - z.max(1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(1, keepdims=True)


def cce_loss(y, p, denom=None):
    """Keras CategoricalCrossentropy(from_logits=False): p/=sum p; clip [1e-7,1-1e-7];
    -sum_c y log p; mean over batch (``denom`` = global batch size when sharded)."""
    p = p / p.sum(1, keepdims=True)
    p = np.clip(p, 1e-7, 1 - 1e-7)
    per = -(y * np.log(p)).sum(1)
    return per.sum() / (len(y) if denom is None else denom)


def softmax_cce_grad(y, p, denom=None):
    """dL/dlogits of softmax followed by the *clipped* Keras CCE:
    dz_k = (p_k * sum_c(y_c m_c) - y_k m_k) / B with m_c = [1e-7 < p_c < 1-1e-7]
    (clip_by_value passes no gradient outside its range).  In the unclipped region this is
    the familiar (p - y)/B of SURVEY 8.A.6."""
    bsz = len(y) if denom is None else denom
    q = p / p.sum(1, keepdims=True)
    m = ((q > 1e-7) & (q < 1 - 1e-7)).astype(p.dtype)
    ym = y * m
    return (p * ym.sum(1, keepdims=True) - ym) / bsz


def cce_loss_from_logits(y, z, denom=None):
    """What Keras' CategoricalCrossentropy() (from_logits=False, gcn.py:326) evaluates INSIDE tf.function
    (train_step, gcn.py:328-335): keras.backend.categorical_crossentropy sees a graph tensor produced by a Softmax op
    (TF >= 2.6: an output carrying _keras_logits), takes that op's input and calls
    tf.nn.softmax_cross_entropy_with_logits -- no renormalisation, no clip:
    loss_g = sum_c y_c * (logsumexp(z) - z_c), mean over the batch."""
    m = z.max(1, keepdims=True)
    lse = m + np.log(np.exp(z - m).sum(1, keepdims=True))
    per = (y * (lse - z)).sum(1)
    return per.sum() / (len(y) if denom is None else denom)


def softmax_cce_grad_from_logits(y, p, denom=None):
    """dL/dlogits of softmax_cross_entropy_with_logits: (p * sum_c y_c - y) / B everywhere."""
    bsz = len(y) if denom is None else denom
    return (p * y.sum(1, keepdims=True) - y) / bsz


def cce(y, logits, probs, denom=None, mode="logits"):
    """(loss, dlogits) of the two Keras code paths: mode "logits" = inside tf.function (train_step), "probs" = on eager
    tensors (evaluate(), gcn.py:351-354, TF < 2.6)."""
    if mode == "logits":
        return cce_loss_from_logits(y, logits, denom), softmax_cce_grad_from_logits(y, probs, denom)
    if mode == "probs":
        return cce_loss(y, probs, denom), softmax_cce_grad(y, probs, denom)
    raise ValueError(mode)


def categorical_accuracy(y, p):
    return float(np.mean(np.argmax(y, 1) == np.argmax(p, 1)))


# --------------------------------------------------------------------------- #
# The M1 benchmark model:
#   GCNConv(F->H,relu) -> GCNConv(H->H,relu) -> GlobalSumPool -> Dense(H->C) softmax
#   CCE loss, SGD.   Step order follows train_step, src/scripts/gcn.py:330-340.
# --------------------------------------------------------------------------- #

def glorot_uniform(rng, fan_in, fan_out, dtype=np.float64):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


def gcn2_init(rng, f_in, hidden, n_classes, dtype=np.float64):
    return {
        "w1": glorot_uniform(rng, f_in, hidden, dtype), "b1": np.zeros(hidden, dtype),
        "w2": glorot_uniform(rng, hidden, hidden, dtype), "b2": np.zeros(hidden, dtype),
        "w3": glorot_uniform(rng, hidden, n_classes, dtype), "b3": np.zeros(n_classes, dtype),
    }


GCN2_PARAM_ORDER = ("w1", "b1", "w2", "b2", "w3", "b3")


def gcn2_forward(params, x, csr, graph_ptr, pool="sum"):
    y1, c1 = gcn_conv_fwd(x, csr, params["w1"], params["b1"], "relu")
    y2, c2 = gcn_conv_fwd(y1, csr, params["w2"], params["b2"], "relu")
    pooled, arg = global_pool_fwd(y2, graph_ptr, pool)
    logits = pooled @ params["w3"] + params["b3"]
    probs = softmax(logits)
    return probs, {"c1": c1, "c2": c2, "pooled": pooled, "arg": arg, "logits": logits,
                   "y1": y1, "y2": y2}


def gcn2_loss_and_grads(params, x, csr, graph_ptr, y, pool="sum", denom=None, csr_t=None, cce_mode="logits"):
    """Returns loss, acc, grads dict, cache.  ``denom`` = global batch size B when this
    call sees only a shard of the batch: summed shard grads == full-batch grads.
    cce_mode "logits" (default) is the loss train_step computes under tf.function; "probs" the eager form."""
    probs, cache = gcn2_forward(params, x, csr, graph_ptr, pool)
    yf = y.astype(probs.dtype)
    loss, dlogits = cce(yf, cache["logits"], probs, denom, cce_mode)
    acc = categorical_accuracy(yf, probs)
    g = {}
    g["w3"] = cache["pooled"].T @ dlogits
    g["b3"] = dlogits.sum(0)
    dpooled = dlogits @ params["w3"].T
    dy2 = global_pool_bwd(dpooled, graph_ptr, x.shape[0], pool, cache["arg"])
    dy1, g["w2"], g["b2"] = gcn_conv_bwd(dy2, cache["c2"], csr, params["w2"], "relu", True, csr_t)
    _, g["w1"], g["b1"] = gcn_conv_bwd(dy1, cache["c1"], csr, params["w1"], "relu", False, csr_t)
    cache["probs"] = probs
    cache["dy2"] = dy2
    cache["dy1"] = dy1
    return loss, acc, g, cache


def sgd_apply(params, grads, lr):
    """Keras SGD, no momentum: w <- w - lr*g  (gcn.py:325,338)."""
    return {k: params[k] - lr * grads[k] for k in params}


def piecewise_lr(step, epochs):
    """PiecewiseConstantDecay(boundaries=[0, floor(.3*epochs)], values=[.02,.002,.0002])
    indexed by optimizer step (gcn.py:321-324)."""
    b0, b1 = 0, int(np.floor(0.3 * epochs))
    if step <= b0:
        return 0.02
    if step <= b1:
        return 0.002
    return 0.0002


# --------------------------------------------------------------------------- #
# n1 tier: Keras BatchNormalization / PReLU / Dense / MLP / GeneralConv / GeneralGNN
# (A.3-A.5).  Weight order per layer follows Keras: kernel, bias, gamma, beta,
# moving_mean, moving_var, alpha.
# --------------------------------------------------------------------------- #

BN_EPS = 1e-3
BN_MOMENTUM = 0.99


def bn_fwd(x, gamma, beta, mov_mean, mov_var, training, count=None, sums=None):
    """Keras BatchNormalization(axis=-1, momentum=.99, eps=1e-3).
    training: batch mean and *biased* variance over rows; moving <- .99 moving + .01 batch.
    ``sums`` = (sum_x[F], sum_x2[F], count) lets a sharded caller pass globally
    all-reduced statistics (SURVEY 8(e) 'If BN is included')."""
    if training:
        if sums is None:
            mean = x.mean(0)
            var = x.var(0)
        else:
            sx, sx2, cnt = sums
            mean = sx / cnt
            var = sx2 / cnt - mean * mean
        new_mm = BN_MOMENTUM * mov_mean + (1 - BN_MOMENTUM) * mean
        new_mv = BN_MOMENTUM * mov_var + (1 - BN_MOMENTUM) * var
    else:
        mean, var = mov_mean, mov_var
        new_mm, new_mv = mov_mean, mov_var
    inv = 1.0 / np.sqrt(var + BN_EPS)
    xhat = (x - mean) * inv
    return gamma * xhat + beta, {"xhat": xhat, "inv": inv, "training": training}, new_mm, new_mv


def bn_bwd(dy, cache, gamma, count=None, red=None):
    """Training-mode BN backward (stats depend on x).  ``red`` = globally reduced
    (sum dy, sum dy*xhat, count) for sharded callers."""
    xhat, inv = cache["xhat"], cache["inv"]
    dgamma_local = (dy * xhat).sum(0)
    dbeta_local = dy.sum(0)
    if not cache["training"]:
        return dy * gamma * inv, dgamma_local, dbeta_local
    if red is None:
        s1, s2, n = dbeta_local, dgamma_local, dy.shape[0]
    else:
        s1, s2, n = red
    dx = (gamma * inv) * (dy - s1 / n - xhat * (s2 / n))
    return dx, dgamma_local, dbeta_local


def dense_bn_act_fwd(x, p, training, act, final_softmax=False, drop=None):
    """One MLP layer / the dense half of GeneralConv: Dense -> BN -> Dropout -> act (A.3: MLP is, per layer,
    Dense -> BatchNormalization -> Dropout(rate) -> PReLU | Activation(final); GeneralConv.call the same chain, A.4).
    A layer without "gamma" has no BatchNormalization (batch_norm=False).  ``drop`` = the Dropout layer's factor per
    element, keep / (1 - rate) (training only; None = Dropout(0) or inference): it multiplies the activation's input, and
    cache["zb"] holds that input (for the last layer: the logits the loss takes)."""
    z = x @ p["kernel"] + p["bias"]
    if "gamma" in p:
        zb, bnc, mm, mv = bn_fwd(z, p["gamma"], p["beta"], p["moving_mean"], p["moving_var"], training)
    else:
        zb, bnc, mm, mv = z, None, None, None
    if drop is not None:
        zb = zb * drop
    if final_softmax:
        y = softmax(zb)                                  # zb = the Softmax op's input (the logits Keras' loss takes)
    elif act == "prelu":
        y = act_fwd(zb, "prelu", p["alpha"])
    else:
        y = act_fwd(zb, act)
    return y, {"x": x, "zb": zb, "bn": bnc, "y": y, "drop": drop}, mm, mv


def dense_bn_act_bwd(dy, cache, p, act, need_dx=True, dzb=None, pos=None):
    """Returns dx and grads {kernel,bias,gamma,beta,alpha?}.  ``dzb`` overrides the
    activation backward (used for the fused softmax+CCE head).
    ``pos`` (bool, shape of zb; test infrastructure): which side of the PReLU / ReLU kink every pre-activation is taken to
    be on, instead of zb > 0.  The gradient of a piecewise-linear activation is discontinuous at zb = 0; an fp32 evaluation
    whose zb differs from this fp64 one by rounding lands on the other side for the few entries within rounding of zero,
    and each such entry moves a gradient by a whole term.  Comparing a device run with this backward evaluated on the
    DEVICE's side of every kink separates arithmetic error from that kink noise (the flips themselves are bounded by the
    caller: a flipped entry must lie within the precision's reach of zero)."""
    g = {}
    zb = cache["zb"]
    if dzb is None:
        if pos is None:
            pos = zb > 0
        if act == "prelu":
            g["alpha"] = (dy * np.where(pos, 0.0, zb)).sum(0)
            dzb = dy * np.where(pos, 1.0, p["alpha"]).astype(dy.dtype)
        elif act == "relu":
            dzb = dy * pos
        else:
            dzb = act_bwd(dy, zb, act)
    if cache.get("drop") is not None:
        dzb = dzb * cache["drop"]                        # Dropout backward: the same factor
    if cache["bn"] is not None:
        dz, g["gamma"], g["beta"] = bn_bwd(dzb, cache["bn"], p["gamma"])
    else:
        dz = dzb
    g["kernel"] = cache["x"].T @ dz
    g["bias"] = dz.sum(0)
    dx = dz @ p["kernel"].T if need_dx else None
    return dx, g


def layer_init(rng, fan_in, fan_out, prelu=True, dtype=np.float64, batch_norm=True):
    p = {"kernel": glorot_uniform(rng, fan_in, fan_out, dtype), "bias": np.zeros(fan_out, dtype)}
    if batch_norm:
        p.update({"gamma": np.ones(fan_out, dtype), "beta": np.zeros(fan_out, dtype),
                  "moving_mean": np.zeros(fan_out, dtype), "moving_var": np.ones(fan_out, dtype)})
    if prelu:
        p["alpha"] = np.zeros(fan_out, dtype)
    return p


def general_gnn_init(rng, f_in, n_out, hidden=256, message_passing=4, pre=2, post=2, dtype=np.float64, connectivity="cat",
                     batch_norm=True, hidden_activation="prelu"):
    """GeneralGNN(output, hidden=256, message_passing=4, pre_process=2, post_process=2,
    connectivity='cat', batch_norm=True, aggregate='sum', hidden_activation='prelu',
    pool='sum')  -- defaults of gcn.py:320 (A.3).  Options (r3): connectivity "sum" (out = z + out: every GeneralConv sees
    `hidden` inputs), batch_norm False (no BatchNormalization layers), hidden_activation "relu" / None (no alpha)."""
    layers = {"pre": [], "gnn": [], "post": []}
    prelu = hidden_activation == "prelu"
    w = f_in
    for _ in range(pre):
        layers["pre"].append(layer_init(rng, w, hidden, prelu, dtype, batch_norm)); w = hidden
    for _ in range(message_passing):
        layers["gnn"].append(layer_init(rng, w, hidden, prelu, dtype, batch_norm))
        w = w + hidden if connectivity == "cat" else hidden
    for k in range(post):
        last = k == post - 1
        layers["post"].append(layer_init(rng, w, n_out if last else hidden, prelu and not last, dtype, batch_norm))
        w = hidden
    return layers


def aggregate_vals(rowptr, aggregate, dtype=np.float64):
    """GeneralConv(aggregate=...) (A.4): "sum" -> None (0 / 1 pattern); "mean" -> 1 / (entries of the target row) per entry
    (tf.math.unsorted_segment_mean over the messages of a row; a row without entries aggregates to 0)."""
    if aggregate == "sum":
        return None
    if aggregate != "mean":
        raise ValueError(f"aggregate={aggregate!r}: 'sum' and 'mean' are restated")
    deg = np.diff(np.asarray(rowptr)).astype(dtype)
    return np.repeat(np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0).astype(dtype), np.diff(np.asarray(rowptr)))


F32_MAX = float(np.finfo(np.float32).max)


def aggregate_minmax(rowptr, colidx, h, mode):
    """GeneralConv(aggregate="max" | "min") (A.4): tf.math.unsorted_segment_max / _min over the messages h[a.indices[:,1]] of
    every target row; a row without messages gets the lowest / largest float32 (TensorFlow's value for an empty segment).
    Returns (out, cnt): cnt = how many messages attain the extremum (the gradient's divisor)."""
    n = len(rowptr) - 1
    red = np.max if mode == "max" else np.min
    out = np.full((n, h.shape[1]), -F32_MAX if mode == "max" else F32_MAX, h.dtype)
    cnt = np.zeros((n, h.shape[1]), h.dtype)
    for t in range(n):
        idx = colidx[rowptr[t]:rowptr[t + 1]]
        if len(idx):
            m = h[idx]
            out[t] = red(m, axis=0)
            cnt[t] = (m == out[t]).sum(0)
    return out, cnt


def aggregate_minmax_bwd(rowptr, colidx, h, out, cnt, dy):
    """TensorFlow's _UnsortedSegmentMinOrMaxGrad: the messages equal to the segment's extremum share its gradient equally."""
    dh = np.zeros_like(h)
    for t in range(len(rowptr) - 1):
        idx = colidx[rowptr[t]:rowptr[t + 1]]
        if len(idx):
            sel = (h[idx] == out[t]).astype(h.dtype)
            np.add.at(dh, idx, sel * (dy[t] / cnt[t]))
    return dh


def aggregate_prod(rowptr, colidx, h):
    """GeneralConv(aggregate="prod") (A.4): tf.math.unsorted_segment_prod over the messages of every target row; a row without
    messages gets 1.  Returns (out, aux): aux = the product of the row's non-zero messages where at most one message is zero
    (= out where none is), 0 where two or more are -- what TensorFlow's _UnsortedSegmentProdGrad works from."""
    n = len(rowptr) - 1
    out = np.ones((n, h.shape[1]), h.dtype)
    aux = np.ones((n, h.shape[1]), h.dtype)
    for t in range(n):
        idx = colidx[rowptr[t]:rowptr[t + 1]]
        if len(idx):
            m = h[idx]
            zero = m == 0
            out[t] = np.prod(m, axis=0)
            aux[t] = np.where(zero.sum(0) >= 2, 0.0, np.prod(np.where(zero, 1.0, m), axis=0))
    return out, aux


def aggregate_prod_bwd(rowptr, colidx, h, out, aux, dy):
    """_UnsortedSegmentProdGrad: grad is zeroed where a segment holds more than one zero; the partial derivative of the product wrt
    a message is prod / message for a non-zero message and the product of the non-zero messages for a zero one."""
    dh = np.zeros_like(h)
    for t in range(len(rowptr) - 1):
        idx = colidx[rowptr[t]:rowptr[t + 1]]
        if len(idx):
            m = h[idx]
            zero = m == 0
            part = np.where(zero, aux[t], out[t] / np.where(zero, 1.0, m))
            np.add.at(dh, idx, part * dy[t])
    return dh


def general_gnn_forward(layers, x, csr, graph_ptr, training, final_activation="softmax", aggregate="sum", pool="sum",
                        connectivity="cat", hidden_activation="prelu", drops=None):
    """A.3: pre-MLP -> 4x [z=GeneralConv(out); out=concat([z,out])] -> global pool -> post-MLP.
    GeneralConv (A.4): h = PReLU(BN(x W + b)); out[t] = sum (or mean) over {(t,s) in a.indices} of h[s]
    (values unused, no self-loop added, no normalisation).  pool: "sum" (gcn.py:320's default), "avg", "max".
    connectivity "sum": out = z + out.  hidden_activation: "prelu" | "relu" | None.  drops: the Dropout layers' factors
    (keep / (1 - rate), one array per layer, {"pre": [...], "gnn": [...], "post": [...]}; training only).
    Returns probs, caches, list of (moving_mean, moving_var) updates in layer order."""
    rowptr, colidx, _ = csr
    minmax = aggregate in ("max", "min", "prod")
    agg = None if minmax else aggregate_vals(rowptr, aggregate, x.dtype)
    caches = {"pre": [], "gnn": [], "post": []}
    stats = []
    act = hidden_activation
    drop = (lambda grp, k: drops[grp][k]) if (drops is not None and training) else (lambda grp, k: None)
    out = x
    for k, p in enumerate(layers["pre"]):
        out, c, mm, mv = dense_bn_act_fwd(out, p, training, act, drop=drop("pre", k)); caches["pre"].append(c); stats.append((mm, mv))
    for k, p in enumerate(layers["gnn"]):
        h, c, mm, mv = dense_bn_act_fwd(out, p, training, act, drop=drop("gnn", k)); stats.append((mm, mv))
        if minmax:
            z, c["agg_cnt"] = aggregate_prod(rowptr, colidx, h) if aggregate == "prod" else aggregate_minmax(rowptr, colidx, h, aggregate)
            c["agg_out"] = z
        else:
            z = spmm_csr(rowptr, colidx, agg, h)
        c["width_in"] = out.shape[1]
        caches["gnn"].append(c)
        out = np.concatenate([z, out], axis=1) if connectivity == "cat" else z + out
    pooled, caches["pool_arg"] = global_pool_fwd(out, graph_ptr, pool)
    caches["pooled_in_width"] = out.shape[1]
    out = pooled
    n_post = len(layers["post"])
    for k, p in enumerate(layers["post"]):
        last = k == n_post - 1
        out, c, mm, mv = dense_bn_act_fwd(out, p, training, act if not last else None,
                                          final_softmax=last and final_activation == "softmax", drop=drop("post", k))
        caches["post"].append(c); stats.append((mm, mv))
    return out, caches, stats


def general_gnn_loss_and_grads(layers, x, csr, graph_ptr, y, csr_t=None, cce_mode="logits", aggregate="sum", pool="sum",
                               connectivity="cat", hidden_activation="prelu", drops=None, masks=None):
    """masks (optional; see dense_bn_act_bwd's ``pos``): {"pre": [...], "gnn": [...], "post": [...]} boolean arrays, one per
    layer with a hidden activation (None entries: the layer's own zb > 0) -- the side of every activation kink the backward
    pass is evaluated on."""
    rowptr, colidx, _ = csr
    mask = (lambda grp, k: None) if masks is None else (lambda grp, k: masks[grp][k] if k < len(masks[grp]) else None)
    minmax = aggregate in ("max", "min", "prod")
    agg = None if minmax else aggregate_vals(rowptr, aggregate, x.dtype)
    act = hidden_activation
    probs, caches, stats = general_gnn_forward(layers, x, csr, graph_ptr, True, aggregate=aggregate, pool=pool,
                                               connectivity=connectivity, hidden_activation=hidden_activation, drops=drops)
    yf = y.astype(probs.dtype)
    loss, dlogits = cce(yf, caches["post"][-1]["zb"], probs, None, cce_mode)
    acc = categorical_accuracy(yf, probs)
    grads = {"pre": [None] * len(layers["pre"]), "gnn": [None] * len(layers["gnn"]),
             "post": [None] * len(layers["post"])}
    d = None
    n_post = len(layers["post"])
    for k in reversed(range(n_post)):
        p, c = layers["post"][k], caches["post"][k]
        if k == n_post - 1:
            d, grads["post"][k] = dense_bn_act_bwd(None, c, p, None, True, dzb=dlogits)
        else:
            d, grads["post"][k] = dense_bn_act_bwd(d, c, p, act, True, pos=mask("post", k))
    d = global_pool_bwd(d, graph_ptr, x.shape[0], pool, caches["pool_arg"])
    for k in reversed(range(len(layers["gnn"]))):
        p, c = layers["gnn"][k], caches["gnn"][k]
        hid = p["kernel"].shape[1]
        dz, dskip = (d[:, :hid], d[:, hid:]) if connectivity == "cat" else (d, d)
        if minmax:
            dh = (aggregate_prod_bwd if aggregate == "prod" else aggregate_minmax_bwd)(rowptr, colidx, c["y"], c["agg_out"], c["agg_cnt"], dz)
        elif csr_t is not None and agg is None:
            dh = spmm_csr(csr_t[0], csr_t[1], None, dz)
        else:
            dh = spmm_csr_T(rowptr, colidx, agg, dz)
        dx, grads["gnn"][k] = dense_bn_act_bwd(dh, c, p, act, True, pos=mask("gnn", k))
        d = dx + dskip
    for k in reversed(range(len(layers["pre"]))):
        p, c = layers["pre"][k], caches["pre"][k]
        d, grads["pre"][k] = dense_bn_act_bwd(d, c, p, act, k > 0, pos=mask("pre", k))
    return loss, acc, grads, probs, stats
