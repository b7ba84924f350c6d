"""Device context, device arrays and the thin op layer over the C ABI (include/gcnx.h)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L


class _Side:
    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.ctx._ck(self.ctx.lib.gcnx_side_begin(self.ctx.h))
        return self

    def __exit__(self, *exc):
        self.ctx._ck(self.ctx.lib.gcnx_side_end(self.ctx.h))
        return False


_DEFAULT_CTX = None


def default_context():
    """The process-wide Context the models use when none is passed (``GeneralGNN(n_labels, activation="softmax")`` -- Spektral's
    own signature, gcn.py:320): device LOCAL_RANK (one process per GPU), created on first use; raises without a GPU."""
    global _DEFAULT_CTX
    if _DEFAULT_CTX is None or not _DEFAULT_CTX._live:
        _DEFAULT_CTX = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _DEFAULT_CTX


def with_default_context(init):
    """Constructor decorator: the leading ``ctx`` may be left out (then it is default_context())."""
    import functools

    @functools.wraps(init)
    def wrapper(self, *args, **kw):
        if "ctx" in kw or (args and isinstance(args[0], Context)):
            return init(self, *args, **kw)
        return init(self, default_context(), *args, **kw)
    return wrapper


class Context:
    """One GPU + one HIP stream (gcnx_ctx).  Not thread-safe; one per process per GPU."""

    def __init__(self, device=0):
        self.lib = L.load()
        h = C.c_void_p()
        L.check(self.lib.gcnx_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self._live = True
        self._force_plan = os.environ.get("GCNX_SPMM_KERNEL", "")[:1] in ("t", "p")

    # -- info ------------------------------------------------------------------------------
    def info(self):
        name = C.create_string_buffer(64)
        cus = C.c_int()
        hbm = C.c_size_t()
        self._ck(self.lib.gcnx_device_info(self.h, name, 64, C.byref(cus), C.byref(hbm)))
        return {"arch": name.value.decode(), "cus": cus.value, "hbm_bytes": hbm.value}

    def _ck(self, rc):
        L.check(rc, self.h)

    SPMM_KERNELS = {"auto": 0, "rows": 1, "tile": 2, "pipe": 3}

    def set_lr_source(self, scalar):
        """The update launches read the learning rate from this 1-element fp32 device array instead of their ``lr`` argument
        (None: the argument again); gcnx_set_lr_source.  A captured step then serves every value of a schedule."""
        self._ck(self.lib.gcnx_set_lr_source(self.h, scalar.ptr if scalar is not None else None))

    def set_tuning(self, key, value):
        """Kernel-selection knobs of this context (gcnx_set_tuning; diagnostics: results never depend on them).
        ``set_tuning("spmm_kernel", "auto" | "rows" | "tile" | "pipe")``, ``("gemm_stream", 0 | 1)``, ..."""
        if key == "spmm_kernel":
            self._force_plan = value in ("tile", "pipe")
            value = self.SPMM_KERNELS[value]
        self._ck(self.lib.gcnx_set_tuning(self.h, key.encode(), int(value)))

    # -- memory ----------------------------------------------------------------------------
    def empty(self, shape, dtype=np.float32):
        return DeviceArray.alloc(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        a = DeviceArray.alloc(self, shape, dtype)
        a.fill_zero()
        return a

    def to_device(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        a = DeviceArray.alloc(self, host.shape, host.dtype)
        if host.nbytes:
            self._ck(self.lib.gcnx_h2d(self.h, a.ptr, host.ctypes.data, host.nbytes))
        return a

    def sync(self):
        self._ck(self.lib.gcnx_sync(self.h))

    def side(self):
        """``with ctx.side(): ...`` -- the gcnx calls inside run on the side stream, after everything submitted
        so far and concurrently with what follows; ``ctx.join()`` makes the main stream wait for them."""
        return _Side(self)

    def join(self):
        self._ck(self.lib.gcnx_side_join(self.h))

    # -- events / graphs -------------------------------------------------------------------
    def event(self):
        return Event(self)

    def capture(self, fn):
        """Capture the gcnx calls made by fn() into a replayable HIP graph."""
        self._ck(self.lib.gcnx_capture_begin(self.h))
        try:
            fn()
        except Exception:
            g = C.c_void_p()
            self.lib.gcnx_capture_end(self.h, C.byref(g))
            if g:
                self.lib.gcnx_graph_destroy(self.h, g)
            raise
        g = C.c_void_p()
        self._ck(self.lib.gcnx_capture_end(self.h, C.byref(g)))
        return Graph(self, g)

    def close(self):
        if self._live:
            self._live = False
            self.lib.gcnx_ctx_destroy(self.h)

    def __del__(self):
        pass  # explicit close(); device arrays may outlive an implicit destructor order


class Event:
    def __init__(self, ctx):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx._ck(ctx.lib.gcnx_event_create(ctx.h, C.byref(self.h)))

    def record(self):
        self.ctx._ck(self.ctx.lib.gcnx_event_record(self.ctx.h, self.h))
        return self

    def elapsed_ms_since(self, start):
        ms = C.c_float()
        self.ctx._ck(self.ctx.lib.gcnx_event_elapsed_ms(self.ctx.h, start.h, self.h, C.byref(ms)))
        return ms.value


class Graph:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def launch(self):
        self.ctx._ck(self.ctx.lib.gcnx_graph_launch(self.ctx.h, self.h))

    def destroy(self):
        if self.h:
            self.ctx.lib.gcnx_graph_destroy(self.ctx.h, self.h)
            self.h = None


class DeviceArray:
    """Row-major device buffer (1-D or 2-D) or a column-slice view of one (ld = row stride)."""

    __slots__ = ("ctx", "ptr", "shape", "dtype", "ld", "_base", "_owner", "_spmm_plan")

    def __init__(self, ctx, ptr, shape, dtype, ld=None, base=None, owner=False):
        self.ctx, self.ptr, self.shape, self.dtype = ctx, ptr, tuple(int(s) for s in shape), np.dtype(dtype)
        self.ld = int(ld) if ld is not None else (self.shape[-1] if self.shape else 1)
        self._base, self._owner = base, owner
        self._spmm_plan = None

    @classmethod
    def _view(cls, base, nelem_off, shape):
        """Contiguous view of ``base`` starting at element ``nelem_off`` with a shape of plain ints -- flat() without its
        checks, for callers that build many views per step (the per-batch activation views of a streamed epoch)."""
        self = object.__new__(cls)
        self.ctx, self.ptr, self.shape, self.dtype = base.ctx, base.ptr + nelem_off * base.dtype.itemsize, shape, base.dtype
        self.ld = shape[-1]
        self._base, self._owner, self._spmm_plan = base, False, None
        return self

    @classmethod
    def alloc(cls, ctx, shape, dtype):
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        ctx._ck(ctx.lib.gcnx_malloc(ctx.h, max(nbytes, 16), C.byref(p)))
        return cls(ctx, p.value, shape, dtype, owner=True)

    @property
    def size(self):
        n = 1
        for d in self.shape:
            n *= d
        return n

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def contiguous(self):
        return len(self.shape) < 2 or self.ld == self.shape[1]

    def cols(self, c0, c1):
        """View of columns [c0, c1) -- Spektral's concat-skip is written in place through these."""
        assert len(self.shape) == 2 and 0 <= c0 <= c1 <= self.shape[1]
        return DeviceArray(self.ctx, self.ptr + c0 * self.dtype.itemsize, (self.shape[0], c1 - c0), self.dtype,
                           ld=self.ld, base=self)

    def flat(self, off, n, shape=None):
        """View of n elements starting at element `off` of a contiguous buffer."""
        assert self.contiguous and off + n <= self.size
        return DeviceArray(self.ctx, self.ptr + off * self.dtype.itemsize, shape or (n,), self.dtype, base=self)

    def fill_zero(self):
        assert self.contiguous
        self.ctx._ck(self.ctx.lib.gcnx_memset(self.ctx.h, self.ptr, 0, self.nbytes))

    def copy_from_host(self, host, wait=True):
        """wait=False: small copies are staged and queued in stream order without stalling the host (gcnx_h2d_async)."""
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.size == self.size and self.contiguous
        fn = self.ctx.lib.gcnx_h2d if wait else self.ctx.lib.gcnx_h2d_async
        self.ctx._ck(fn(self.ctx.h, self.ptr, host.ctypes.data, host.nbytes))

    def numpy(self):
        if self.contiguous:
            out = np.empty(self.shape, dtype=self.dtype)
            if out.nbytes:
                self.ctx._ck(self.ctx.lib.gcnx_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
            return out
        # strided view: fetch the enclosing rows and slice on the host
        rows, cols = self.shape
        span = (rows - 1) * self.ld + cols
        buf = np.empty(span, dtype=self.dtype)
        self.ctx._ck(self.ctx.lib.gcnx_d2h(self.ctx.h, buf.ctypes.data, self.ptr, buf.nbytes))
        full = np.zeros(rows * self.ld, dtype=self.dtype)
        full[:span] = buf
        return full.reshape(rows, self.ld)[:, :cols].copy()

    def free(self):
        if self._owner and self.ptr and self.ctx._live:
            self.ctx.lib.gcnx_free(self.ctx.h, self.ptr)
        self.ptr = None
        self._owner = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceCSR:
    """Adjacency of a disjoint batch on the device: CSR int32 (+ fp32 values or None = ones).

    ``block_ptr`` (= graph_ptr) marks the diagonal blocks; ``symmetric`` lets the backward pass
    reuse this CSR for A^T (true for the reference's undirected contact graphs and for
    gcn_filter of a symmetric A); otherwise ``transpose()`` builds the transposed CSR once.
    ``symmetric=None`` (the default of the constructors that take host data): checked once on the device
    (gcnx_csr_inspect) -- the reference's loader accepts any scipy matrix, and a directed adjacency run through the
    symmetric shortcut would give silently wrong gradients.  The same pass checks that no entry leaves its
    graph_ptr block (what the tile plan and the folded pool backward assume); a batch that fails it is refused.
    """

    def __init__(self, ctx, n, nnz, rowptr, colidx, vals=None, block_ptr=None, n_blocks=0, symmetric=None, max_block_rows=0):
        self.ctx, self.n, self.nnz = ctx, int(n), int(nnz)
        self.max_block_rows = int(max_block_rows)     # rows of the largest graph, where the constructor knew graph_ptr on the host (0: unknown)
        self.rowptr, self.colidx, self.vals = rowptr, colidx, vals
        self.block_ptr, self.n_blocks = block_ptr, int(n_blocks)
        self._t = None
        self._plan = None
        self.dense_shape = (self.n, self.n)
        if symmetric is None:
            props = self.inspect()
            symmetric = bool(props & L.CSR_SYMMETRIC)
            if block_ptr is not None and self.n_blocks > 0 and not props & L.CSR_BLOCK_DIAGONAL:
                raise ValueError("adjacency is not block-diagonal with respect to graph_ptr (a DisjointLoader batch is: "
                                 "sp.block_diag); graph_ptr must start at 0, end at N and be non-decreasing")
        self.symmetric = bool(symmetric)

    def inspect(self):
        """Device-side structure check (gcnx_csr_inspect): bit set of L.CSR_SYMMETRIC / CSR_BLOCK_DIAGONAL /
        CSR_GRAPH_PTR_OK."""
        props = C.c_int(0)
        has_blocks = self.block_ptr is not None and self.n_blocks > 0
        self.ctx._ck(self.ctx.lib.gcnx_csr_inspect(self.ctx.h, self.rowptr.ptr, self.colidx.ptr,
                                                    self.vals.ptr if self.vals is not None else None, self.n,
                                                    self.block_ptr.ptr if has_blocks else None,
                                                    self.n_blocks if has_blocks else 0, C.byref(props)))
        return props.value

    @property
    def plan(self):
        """gcnx_spmm_plan of this batch's block structure (built on first use, shared by every
        CSR that shares block_ptr: normalised / unweighted / transposed views)."""
        if self.block_ptr is None or self.n_blocks == 0:
            return None
        # The tile kernels need >= 4 (graph, 32-column slab) units per CU (gcnx_spmm_csr); below ~128 graphs no
        # width reaches that, and building a plan (a D2H copy, a host sort, an upload) per streamed batch is wasted.
        # (r3: a large batch of FEW graphs -- config 5: 122 power-law graphs of 8 192 nodes -- gets a plan too: it lists the hub
        # rows, which then run as segments on workgroups of their own)
        # (r4: and a batch that holds a graph of >= 4096 rows, whatever its size -- the plan lists the column-block work of such
        # graphs, whose feature rows do not fit an XCD's L2: spmm_cb_kernel)
        if self.n_blocks < 128 and self.n < 131072 and self.max_block_rows < 4096 and not getattr(self.ctx, "_force_plan", False):
            return None
        holder = self.block_ptr
        p = getattr(holder, "_spmm_plan", None)
        if p is None:
            h = C.c_void_p()
            self.ctx._ck(self.ctx.lib.gcnx_spmm_plan_create(self.ctx.h, holder.ptr, self.n_blocks, C.byref(h)))
            p = _Plan(self.ctx, h)
            holder._spmm_plan = p
        if self._plan is not p:
            # the tile kernels' degree order of THIS operator's rows (gcnx_spmm_plan_bind: synchronises and rebuilds, so it
            # happens here, ONCE per (plan, rowptr) -- the normalised / unweighted / row-mean views share rowptr and with it
            # the order -- and never inside a captured call: a view created inside a captured sequence finds the order bound)
            key = (self.rowptr.ptr, self.n)
            if key not in p.bound:
                self.ctx._ck(self.ctx.lib.gcnx_spmm_plan_bind(self.ctx.h, p.h, self.rowptr.ptr, self.n))
                p.bound.add(key)
            self._plan = p
        return p.h

    def rebind(self):
        """The row pointers behind ``rowptr`` were rewritten in place: rebuild the plan's row order for them
        (gcnx_spmm_plan_bind, case (b) of include/gcnx.h).  Not inside a captured sequence."""
        self._plan = None
        p = getattr(self.block_ptr, "_spmm_plan", None) if self.block_ptr is not None else None
        if p is not None:
            p.bound.discard((self.rowptr.ptr, self.n))
        return self.plan

    @classmethod
    def from_host_csr(cls, ctx, rowptr, colidx, vals=None, graph_ptr=None, symmetric=None):
        n = len(rowptr) - 1
        d_rp = ctx.to_device(rowptr, np.int32)
        d_ci = ctx.to_device(colidx, np.int32)
        d_v = ctx.to_device(vals, np.float32) if vals is not None else None
        d_gp = ctx.to_device(graph_ptr, np.int32) if graph_ptr is not None else None
        return cls(ctx, n, len(colidx), d_rp, d_ci, d_v, d_gp, 0 if graph_ptr is None else len(graph_ptr) - 1, symmetric,
                   _max_block(graph_ptr))

    @classmethod
    def from_coo(cls, ctx, indices, values, n, graph_ptr=None, symmetric=None, weighted=True):
        """DisjointLoader's SparseTensor (indices[nnz,2] int64 row-major sorted) -> device CSR
        via gcnx_coo_to_csr (the COO never becomes a host CSR)."""
        indices = np.asarray(indices)
        nnz = indices.shape[0]
        rows = ctx.to_device(indices[:, 0], np.int64)
        cols = ctx.to_device(indices[:, 1], np.int64)
        d_rp = ctx.empty(n + 1, np.int32)
        d_ci = ctx.empty(max(nnz, 1), np.int32)
        ctx._ck(ctx.lib.gcnx_coo_to_csr(ctx.h, rows.ptr, cols.ptr, nnz, n, d_rp.ptr, d_ci.ptr))
        rows.free(); cols.free()
        d_v = ctx.to_device(values, np.float32) if (weighted and values is not None) else None
        d_gp = ctx.to_device(graph_ptr, np.int32) if graph_ptr is not None else None
        return cls(ctx, n, nnz, d_rp, d_ci, d_v, d_gp, 0 if graph_ptr is None else len(graph_ptr) - 1, symmetric, _max_block(graph_ptr))

    def gcn_norm(self, mode="spektral"):
        """Device gcn_filter: returns a CSR sharing structure with self, values = A^."""
        out = self.ctx.empty(max(self.nnz, 1), np.float32)
        self.ctx._ck(self.ctx.lib.gcnx_gcn_norm(self.ctx.h, self.rowptr.ptr, self.colidx.ptr,
                                                 self.vals.ptr if self.vals is not None else None, self.n,
                                                 L.NORM_SPEKTRAL if mode == "spektral" else L.NORM_PYG, out.ptr))
        return DeviceCSR(self.ctx, self.n, self.nnz, self.rowptr, self.colidx, out, self.block_ptr, self.n_blocks,
                         self.symmetric, self.max_block_rows)

    def transpose(self):
        if self.symmetric:
            return self
        if self._t is None:
            ctx = self.ctx
            rp = ctx.empty(self.n + 1, np.int32)
            ci = ctx.empty(max(self.nnz, 1), np.int32)
            v = ctx.empty(max(self.nnz, 1), np.float32) if self.vals is not None else None
            ctx._ck(ctx.lib.gcnx_csr_transpose(ctx.h, self.rowptr.ptr, self.colidx.ptr,
                                               self.vals.ptr if self.vals is not None else None, self.n, self.nnz,
                                               rp.ptr, ci.ptr, v.ptr if v is not None else None))
            self._t = DeviceCSR(ctx, self.n, self.nnz, rp, ci, v, self.block_ptr, self.n_blocks, False, self.max_block_rows)
        return self._t

    def row_mean(self):
        """Same structure with 1 / (entries of the row) on every entry: GeneralConv(aggregate="mean") -- the mean over a row's
        messages (tf.math.unsorted_segment_mean); a row without entries aggregates to 0.  Built once per operator."""
        if getattr(self, "_row_mean", None) is None:
            deg = np.diff(self.rowptr.numpy()).astype(np.int64)
            vals = np.repeat(np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0).astype(np.float32), deg)
            self._row_mean = DeviceCSR(self.ctx, self.n, self.nnz, self.rowptr, self.colidx, self.ctx.to_device(vals), self.block_ptr,
                                       self.n_blocks, False, self.max_block_rows)
        return self._row_mean

    def unweighted(self):
        """Same structure, values ignored (GeneralConv aggregation).  One view per operator (as row_mean): the models ask
        for it inside their step sequence, and a view keeps its transposed CSR and its plan binding."""
        if self.vals is None:
            return self
        if getattr(self, "_unweighted", None) is None:
            self._unweighted = DeviceCSR(self.ctx, self.n, self.nnz, self.rowptr, self.colidx, None, self.block_ptr, self.n_blocks,
                                         self.symmetric, self.max_block_rows)
        return self._unweighted


def _max_block(graph_ptr):
    gp = np.asarray(graph_ptr) if graph_ptr is not None else None
    return int(np.diff(gp).max()) if gp is not None and gp.size > 1 else 0


class _Plan:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        self.bound = set()          # (rowptr pointer, n) pairs whose row order the plan holds

    def __del__(self):
        try:
            if self.h and self.ctx._live:
                self.ctx.lib.gcnx_spmm_plan_destroy(self.ctx.h, self.h)
        except Exception:
            pass


class Segments:
    """Graph membership of the rows of a disjoint batch: graph_ptr int32[B+1] on the device
    (the sorted id vector ``i`` of DisjointLoader, run-length encoded)."""

    def __init__(self, ctx, graph_ptr_host):
        gp = np.asarray(graph_ptr_host, dtype=np.int32)
        self.ctx, self.n_graphs, self.n = ctx, len(gp) - 1, int(gp[-1])
        self.host = gp
        self.dev = ctx.to_device(gp, np.int32)

    @classmethod
    def from_device(cls, ctx, dev, graph_ptr_host):
        """graph_ptr already on the device (device-side collate); the host copy is kept for shapes."""
        self = cls.__new__(cls)
        gp = np.asarray(graph_ptr_host, dtype=np.int32)
        self.ctx, self.n_graphs, self.n, self.host, self.dev = ctx, len(gp) - 1, int(gp[-1]), gp, dev
        return self

    @property
    def has_empty(self):
        """True if some graph of the batch has no nodes (never out of DisjointLoader; possible through the raw surface)."""
        if getattr(self, "_has_empty", None) is None:
            self._has_empty = bool(self.n_graphs) and bool(np.any(np.diff(self.host) <= 0))
        return self._has_empty

    @property
    def ids(self):
        """The DisjointLoader id vector i[N] (graph of every row) on the device, built on first use."""
        if getattr(self, "_ids", None) is None:
            self._ids = self.ctx.to_device(np.repeat(np.arange(self.n_graphs, dtype=np.int32), np.diff(self.host)), np.int32)
        return self._ids

    @classmethod
    def from_ids(cls, ctx, i, n_graphs=None):
        i = np.asarray(i)
        b = int(i.max()) + 1 if n_graphs is None and i.size else (n_graphs or 0)
        if i.size and np.any(np.diff(i) < 0):
            raise ValueError("segment ids must be sorted (DisjointLoader order)")
        gp = np.zeros(b + 1, dtype=np.int64)
        np.cumsum(np.bincount(i, minlength=b), out=gp[1:])
        return cls(ctx, gp)


# ------------------------------------------------------------------------------------------- #
# ops: one function per C entry point, operating on DeviceArrays
# ------------------------------------------------------------------------------------------- #

def _p(a):
    """Device pointer of an fp32 operand (or None).  A float64 upload would be silently
    reinterpreted by the kernels, so the dtype is checked here, at the boundary."""
    if a is None:
        return None
    if a.dtype != np.float32 and a.dtype != np.int32:
        raise TypeError(f"libgcnx operands are float32/int32, got {a.dtype}")
    return a.ptr


def gemm(ctx, x, w, bias, out, act=None, prec="f32", alpha=None):
    n, fi = x.shape
    fo = w.shape[1]
    assert w.shape[0] == fi and out.shape == (n, fo)
    ctx._ck(ctx.lib.gcnx_gemm(ctx.h, _p(x), x.ld, _p(w), _p(bias), _p(out), out.ld, n, fi, fo, L.PRECS[prec],
                              L.ACTS[act], _p(alpha)))
    return out


def gemm_relu_bits(ctx, x, w, bias, out, bits, prec="bf16"):
    """out = relu(x w + bias) and the bit image of [out > 0] for gemm_dx(mask_bits=...) (gcnx_gemm_relu_bits).  Returns
    False -- nothing launched -- when the streaming bf16 kernel does not serve the shape: call gemm(act="relu") then."""
    n, fi = x.shape
    fo = w.shape[1]
    assert w.shape[0] == fi and out.shape == (n, fo) and bits.nbytes >= n * 64
    rc = ctx.lib.gcnx_gemm_relu_bits(ctx.h, _p(x), x.ld, _p(w), _p(bias), _p(out), out.ld, n, fi, fo, L.PRECS[prec], bits.ptr)
    if rc == L.ERR_UNSUPPORTED:
        return False
    ctx._ck(rc)
    return True


# ---- bf16 STORAGE between bf16-operand weight GEMMs (include/gcnx.h: "bf16 STORAGE ...").  Arrays of dtype uint16 hold
# bfloat16 bit patterns.  Each wrapper returns False -- nothing launched -- when the library answers GCNX_ERR_UNSUPPORTED.
def _is16(a):
    return a.dtype == np.uint16


def spmm_bf16out(ctx, a, h, bias, out16, act=None):
    """out16 = bf16(act(A h + bias)) (gcnx_spmm_csr_bf16out): the aggregation with its result stored as bfloat16."""
    n, f = h.shape
    assert a.n == n and out16.shape == (n, f) and _is16(out16)
    rc = ctx.lib.gcnx_spmm_csr_bf16out(ctx.h, a.rowptr.ptr, a.colidx.ptr, _p(a.vals), _p(h), h.ld, _p(bias), out16.ptr, out16.ld,
                                       n, f, L.ACTS[act], a.plan)
    if rc == L.ERR_UNSUPPORTED:
        return False
    ctx._ck(rc)
    return True


def spmm_pool_bwd_bf16out(ctx, at, y, seg, dpooled, out16, mode="sum", y_bits=None):
    """out16 = bf16(A^T (pool'(dpooled) * [y > 0])) from the bit image of y (gcnx_spmm_csr_pool_bwd_bf16out)."""
    n, f = y.shape
    assert at.n == n and out16.shape == (n, f) and _is16(out16) and dpooled.shape == (seg.n_graphs, f)
    if y_bits is None or at.n_blocks != seg.n_graphs:
        return False
    rc = ctx.lib.gcnx_spmm_csr_pool_bwd_bf16out(ctx.h, at.rowptr.ptr, at.colidx.ptr, _p(at.vals), _p(y), y.ld, seg.dev.ptr,
                                                seg.n_graphs, _p(dpooled), dpooled.ld, out16.ptr, out16.ld, n, f, L.POOLS[mode],
                                                at.plan, _p(y_bits))
    if rc == L.ERR_UNSUPPORTED:
        return False
    ctx._ck(rc)
    return True


def stream_images(ctx, jobs, storage):
    """The bf16 images of up to four 256 x 256 weight operands in one launch (gcnx_gemm_stream_images).  jobs: (w, transpose)
    pairs -- transpose = True: the operand of gemm_fwd_bf16 (x w), False: of gemm_dx_bf16 (dh w^T); storage: a uint16 array
    of at least len(jobs) * 65536 elements.  Returns the device pointers, to be passed as ``wimg``."""
    assert storage.dtype == np.uint16 and storage.size >= len(jobs) * (L.STREAM_IMAGE_BYTES // 2) and len(jobs) <= 4
    arr = (L.StreamImageJob * len(jobs))()
    ptrs = []
    for i, (w, tr) in enumerate(jobs):
        assert w.shape == (256, 256) and w.contiguous
        ptrs.append(storage.ptr + i * L.STREAM_IMAGE_BYTES)
        arr[i] = L.StreamImageJob(w.ptr, 256, 256, 1 if tr else 0, ptrs[-1])
    ctx._ck(ctx.lib.gcnx_gemm_stream_images(ctx.h, len(jobs), C.cast(arr, C.c_void_p)))
    return ptrs


def gemm_fwd_bf16(ctx, x16, w, bias, out, act=None, bits=None, wimg=None):
    """out = act(x16 w + bias) with x16 stored as bfloat16; out is stored as bfloat16 when it is a uint16 array
    (gcnx_gemm_fwd_bf16); bits (optional, act = "relu"): the bit image of [out > 0] for gemm_dx_bf16."""
    n, fi = x16.shape
    fo = w.shape[1]
    assert _is16(x16) and w.shape[0] == fi and out.shape == (n, fo) and (bits is None or bits.nbytes >= n * 64)
    rc = ctx.lib.gcnx_gemm_fwd_bf16(ctx.h, x16.ptr, x16.ld, _p(w), _p(bias), out.ptr, out.ld, int(_is16(out)), n, fi, fo, L.ACTS[act],
                                    _p(bits), wimg)
    if rc == L.ERR_UNSUPPORTED:
        return False
    ctx._ck(rc)
    return True


def gemm_dx_bf16(ctx, dh16, w, dx, mask_bits=None, db=None, wimg=None):
    """dx = dh16 w^T (* the ReLU mask in mask_bits), db = column sums of dx; dh16 stored as bfloat16, dx as bfloat16 when it
    is a uint16 array (gcnx_gemm_dx_bf16)."""
    n, fo = dh16.shape
    fi = w.shape[0]
    assert _is16(dh16) and w.shape[1] == fo and dx.shape == (n, fi)
    rc = ctx.lib.gcnx_gemm_dx_bf16(ctx.h, dh16.ptr, dh16.ld, _p(w), dx.ptr, dx.ld, int(_is16(dx)), n, fi, fo, _p(mask_bits), _p(db), wimg)
    if rc == L.ERR_UNSUPPORTED:
        return False
    ctx._ck(rc)
    return True


def gemm_dw_bf16(ctx, x16, dh16, dw):
    """dw = x16^T dh16, both stored as bfloat16 (gcnx_gemm_dw_bf16)."""
    n, fi = x16.shape
    fo = dh16.shape[1]
    assert _is16(x16) and _is16(dh16) and dh16.shape[0] == n and dw.shape == (fi, fo)
    rc = ctx.lib.gcnx_gemm_dw_bf16(ctx.h, x16.ptr, x16.ld, dh16.ptr, dh16.ld, _p(dw), n, fi, fo)
    if rc == L.ERR_UNSUPPORTED:
        return False
    ctx._ck(rc)
    return True


def spmm(ctx, a, h, bias, out, act=None):
    n, f = h.shape
    assert a.n == n and out.shape == (n, f)
    ctx._ck(ctx.lib.gcnx_spmm_csr(ctx.h, a.rowptr.ptr, a.colidx.ptr, _p(a.vals), _p(h), h.ld, _p(bias), _p(out),
                                  out.ld, n, f, L.ACTS[act], a.plan))
    return out


def segment_pool(ctx, seg, x, pooled, mode="sum", argmax=None):
    ctx._ck(ctx.lib.gcnx_segment_pool(ctx.h, seg.dev.ptr, _p(x), x.ld, _p(pooled), seg.n_graphs, x.shape[1],
                                      L.POOLS[mode], _p(argmax)))
    return pooled


def segment_pool_bwd(ctx, seg, dpooled, dx, mode="sum", argmax=None, y=None, db=None):
    n, f = dx.shape
    ctx._ck(ctx.lib.gcnx_segment_pool_bwd(ctx.h, seg.dev.ptr, _p(dpooled), _p(dx), dx.ld, n, seg.n_graphs, f,
                                          L.POOLS[mode], _p(argmax), _p(y), y.ld if y is not None else 0, _p(db)))
    return dx


def pool_dense_softmax_cce(ctx, seg, x, pooled, w, bias, y, probs, loss_acc=None, denom=None, dw=None, db=None,
                           dpooled=None, mode="sum", argmax=None, db_relu=None, cce="logits"):
    """segment_pool + dense_softmax_cce as one call (gcnx_pool_dense_softmax_cce): ``pooled`` is written as well;
    ``db_relu`` = column sums of pool'(dpooled) * [x > 0] (the bias gradient of the ReLU layer that produced x)."""
    b, h = pooled.shape
    c = w.shape[1]
    assert x.shape[1] == h and seg.n_graphs == b
    ctx._ck(ctx.lib.gcnx_pool_dense_softmax_cce(ctx.h, seg.dev.ptr, _p(x), x.ld, L.POOLS[mode], _p(argmax), _p(pooled),
                                                pooled.ld, _p(w), _p(bias), _p(y), b, h, c,
                                                float(denom if denom else max(b, 1)), _p(probs), _p(loss_acc), _p(dw),
                                                _p(db), _p(dpooled), dpooled.ld if dpooled is not None else 0,
                                                _p(db_relu), L.CCES[cce]))
    return probs


def pooled_dense_softmax_cce(ctx, seg, pooled, cnt, w, bias, y, probs, loss_acc=None, denom=None, dw=None, db=None, dpooled=None,
                             mode="sum", db_relu=None, cce="logits"):
    """dense_softmax_cce on pooled rows that are already there (spmm_relu_bits_pool), + db_relu from the positive counts
    (gcnx_pooled_dense_softmax_cce)."""
    b, h = pooled.shape
    c = w.shape[1]
    assert seg.n_graphs == b and (cnt is None or cnt.shape == (b, h))
    ctx._ck(ctx.lib.gcnx_pooled_dense_softmax_cce(ctx.h, seg.dev.ptr, L.POOLS[mode], _p(pooled), pooled.ld, _p(cnt), _p(w), _p(bias), _p(y),
                                                  b, h, c, float(denom if denom else max(b, 1)), _p(probs), _p(loss_acc), _p(dw), _p(db),
                                                  _p(dpooled), dpooled.ld if dpooled is not None else 0, _p(db_relu), L.CCES[cce]))
    return probs


def spmm_relu_bits(ctx, a, h, bias, out, bits):
    """out = relu(A h + bias) on the tile kernels, which also write the bit image of [out > 0] (int32 [(f / 32) * n]) that
    spmm_pool_bwd(y_bits=...) folds from.  Returns False -- nothing launched -- if the tile kernels do not serve this
    batch (gcnx_spmm_csr_relu_bits: GCNX_ERR_UNSUPPORTED); the caller then takes spmm()."""
    n, f = h.shape
    assert a.n == n and out.shape == (n, f) and bits.size >= (f // 32) * n
    rc = ctx.lib.gcnx_spmm_csr_relu_bits(ctx.h, a.rowptr.ptr, a.colidx.ptr, _p(a.vals), _p(h), h.ld, _p(bias), _p(out), out.ld,
                                         n, f, a.plan, _p(bits))
    if rc == L.ERR_UNSUPPORTED:
        return False
    ctx._ck(rc)
    return True


def spmm_pool_bwd(ctx, at, y, seg, dpooled, out, mode="sum", y_bits=None):
    """out = A^T (pool'(dpooled) * [y > 0]) in one gather (gcnx_spmm_csr_pool_bwd); ``at`` is the transposed operator."""
    n, f = y.shape
    assert at.n == n and out.shape == (n, f) and dpooled.shape == (seg.n_graphs, f)
    ctx._ck(ctx.lib.gcnx_spmm_csr_pool_bwd(ctx.h, at.rowptr.ptr, at.colidx.ptr, _p(at.vals), _p(y), y.ld, seg.dev.ptr,
                                           seg.n_graphs, _p(dpooled), dpooled.ld, _p(out), out.ld, n, f, L.POOLS[mode],
                                           at.plan if at.n_blocks == seg.n_graphs else None, _p(y_bits)))
    return out


def spmm_relu_bits_pool(ctx, a, h, bias, out, bits, seg, pooled, cnt, mode="sum"):
    """The pooled GCNConv's forward of a training step without its output (gcnx_spmm_csr_relu_bits_pool): the bit image of
    relu(A h + b), the graphs' pooled rows and positive counts; rows of ``out`` are written for graphs taller than a tile only.
    False -- nothing launched -- when the tile kernels do not serve the batch."""
    n, f = h.shape
    assert a.n == n and out.shape == (n, f) and pooled.shape == (seg.n_graphs, f) and cnt.shape == (seg.n_graphs, f)
    if a.plan is None or a.n_blocks != seg.n_graphs:
        return False
    rc = ctx.lib.gcnx_spmm_csr_relu_bits_pool(ctx.h, a.rowptr.ptr, a.colidx.ptr, _p(a.vals), _p(h), h.ld, _p(bias), _p(out), out.ld, n, f,
                                              a.plan, bits.ptr, seg.dev.ptr, seg.n_graphs, L.POOLS[mode], _p(pooled), pooled.ld, _p(cnt))
    if rc == L.ERR_UNSUPPORTED:
        return False
    ctx._ck(rc)
    return True


def gcn_conv_fused_ok(ctx, n, fi, fo, ldx=None):
    """True if gcn_conv_fwd / gcn_conv_bwd_pool serve these shapes (gcnx_gcn_conv_fused_ok)."""
    return bool(ctx.lib.gcnx_gcn_conv_fused_ok(int(n), int(fi), int(fo), int(ldx if ldx is not None else fi)))


def gcn_conv_fwd(ctx, a, x, w, bias, out, act="relu", s=None, wt=None, prec="f32", pool=None):
    """out = act((A x) w + bias) in one launch; s (optional) receives A x, wt (optional, [fo, fi]) w^T
    (gcnx_gcn_conv_fwd).  pool = (seg, tile_part, tile_cnt): the launch also leaves the global pool's per-tile partial
    sums / positive counts of ``out`` in the two [pool_tile_rows(n, b), fo] arrays (gcnx_gcn_conv_fwd_pool)."""
    n, fi = x.shape
    fo = w.shape[1]
    assert a.n == n and w.shape[0] == fi and w.contiguous and out.shape == (n, fo) and (s is None or s.shape == (n, fi))
    if pool is None:
        ctx._ck(ctx.lib.gcnx_gcn_conv_fwd(ctx.h, a.rowptr.ptr, a.colidx.ptr, _p(a.vals), _p(x), x.ld, n, fi, _p(w), fo, _p(bias),
                                          L.ACTS[act], _p(s), s.ld if s is not None else 0, _p(out), out.ld, _p(wt), L.PRECS[prec]))
        return out
    seg, tp, tc = pool
    assert tp.shape == tc.shape == (pool_tile_rows(n, seg.n_graphs), fo) and tp.contiguous and tc.contiguous
    ctx._ck(ctx.lib.gcnx_gcn_conv_fwd_pool(ctx.h, a.rowptr.ptr, a.colidx.ptr, _p(a.vals), _p(x), x.ld, n, fi, _p(w), fo, _p(bias),
                                           L.ACTS[act], _p(s), s.ld if s is not None else 0, _p(out), out.ld, _p(wt), L.PRECS[prec],
                                           seg.ids.ptr, seg.n_graphs, _p(tp), _p(tc)))
    return out


def to_bf16(ctx, x):
    """bf16 copy (uint16 bit patterns, round to nearest even) of a contiguous fp32 array (gcnx_f32_to_bf16)."""
    assert x.contiguous and x.dtype == np.float32
    y = ctx.empty(x.shape, np.uint16)
    ctx._ck(ctx.lib.gcnx_f32_to_bf16(ctx.h, x.ptr, y.ptr, x.size))
    return y


def to_bf16_into(ctx, x, y16):
    """y16 <- bf16(x), round to nearest even (gcnx_f32_to_bf16), into an existing uint16 array of the same shape."""
    assert x.contiguous and y16.contiguous and x.dtype == np.float32 and y16.dtype == np.uint16 and x.shape == y16.shape
    ctx._ck(ctx.lib.gcnx_f32_to_bf16(ctx.h, x.ptr, y16.ptr, x.size))
    return y16


def from_bf16(ctx, x):
    assert x.contiguous and x.dtype == np.uint16
    y = ctx.empty(x.shape, np.float32)
    ctx._ck(ctx.lib.gcnx_bf16_to_f32(ctx.h, x.ptr, y.ptr, x.size))
    return y


def spmm_bf16(ctx, a, h, bias, out, act=None):
    """out = bf16(act(A h + bias)) with h and out stored as bf16 (uint16 arrays), fp32 accumulation (gcnx_spmm_csr_bf16)."""
    n, f = h.shape
    assert a.n == n and out.shape == (n, f) and h.dtype == np.uint16 and out.dtype == np.uint16
    ctx._ck(ctx.lib.gcnx_spmm_csr_bf16(ctx.h, a.rowptr.ptr, a.colidx.ptr, _p(a.vals), h.ptr, h.ld, _p(bias), out.ptr, out.ld, n, f,
                                       L.ACTS[act]))
    return out


def pool_tile_rows(n, b):
    """Rows of the per-tile partial-sum arrays gcn_conv_fwd(pool=...) fills: one per (32-row tile, graph) pair at most."""
    return (int(n) + 31) // 32 + int(b)


def head_args(seg, tile_part, tile_cnt, pool_sum, pool_cnt, w, bias, y, denom, probs, loss_acc, dw, db, db_relu, pooled, dpooled,
              mode="sum", cce="logits"):
    """gcnx_head_args for gcn_conv_bwd_pool(head=...) and gemm_dw2(leaf=...).  The DeviceArrays must outlive the calls."""
    b, h = seg.n_graphs, w.shape[0]
    c = w.shape[1]
    assert w.contiguous and pooled.contiguous and dpooled.contiguous and pooled.shape == (b, h) and dpooled.shape == (b, h)
    assert tile_part.shape[1] == h and tile_cnt.shape == tile_part.shape and tile_part.contiguous and tile_cnt.contiguous
    assert pool_sum.shape == (b, h) and pool_cnt.shape == (b, h) and pool_sum.contiguous and pool_cnt.contiguous
    return L.HeadArgs(_p(tile_part), _p(tile_cnt), tile_part.shape[0], _p(pool_sum), _p(pool_cnt), seg.dev.ptr, b, h, L.POOLS[mode],
                      _p(w), _p(bias), _p(y), c, float(denom), L.CCES[cce], _p(probs), _p(loss_acc), _p(dw), _p(db), _p(db_relu),
                      _p(pooled), _p(dpooled))


def gcn_conv_bwd_pool(ctx, at, y2, seg, dpooled, w2, y1, dz2, dz1, db1=None, mode="sum", scratch=None, w2t=None, prec="f32",
                      head=None):
    """dz2 = pool'(dpooled) * [y2 > 0], dz1 = ((A^T dz2) w2^T) * [y1 > 0], db1 = column sums of dz1 -- one launch
    (gcnx_gcn_conv_bwd_pool).  With ``scratch`` the db1 reduction is left pending: returns the PendingReduce for
    gemm_dw2 (all zeros when nothing is pending).  w2t: w2^T as written by gcn_conv_fwd(wt=...), read instead of w2."""
    n, f2 = y2.shape
    f1 = w2.shape[0]
    assert at.n == n and w2.shape == (f1, f2) and w2.contiguous and y1.shape == (n, f1) and dz1.shape == (n, f1)
    assert (head is not None or dpooled.shape == (seg.n_graphs, f2)) and (dz2 is None or dz2.shape == (n, f2))
    pend = L.PendingReduce()
    ctx._ck(ctx.lib.gcnx_gcn_conv_bwd_pool(ctx.h, at.rowptr.ptr, at.colidx.ptr, _p(at.vals), _p(y2), y2.ld, seg.ids.ptr,
                                           seg.dev.ptr, seg.n_graphs, _p(dpooled), dpooled.ld if dpooled is not None else 0,
                                           L.POOLS[mode], n, f2,
                                           _p(w2t if w2t is not None else w2), f1, 1 if w2t is not None else 0, _p(y1), y1.ld, _p(dz2), dz2.ld if dz2 is not None else 0, _p(dz1), dz1.ld, _p(db1),
                                           _p(scratch), scratch.size if scratch is not None else 0,
                                           C.byref(pend) if scratch is not None else None, L.PRECS[prec],
                                           C.byref(head) if head is not None else None))
    return pend


def gcn_conv_bwd_scratch_floats(ctx, n, f1):
    return int(ctx.lib.gcnx_gcn_conv_bwd_scratch_floats(int(n), int(f1)))


def gemm_dw2(ctx, xa, dha, dwa, xb, dhb, dwb, prec="f32", params=None, grads=None, lr=0.0, pending=None, leaf=None):
    """dwa = xa^T dha and dwb = xb^T dhb in one launch; with params / grads (the flat buffers both gradients are
    views of) the reduction also applies the SGD step and finishes ``pending`` (gcnx_gemm_dw2)."""
    n, fia = xa.shape
    foa = dha.shape[1]
    fib, fob = xb.shape[1], dhb.shape[1]
    assert xb.shape[0] == n and dha.shape[0] == n and dhb.shape[0] == n and dwa.shape == (fia, foa) and dwb.shape == (fib, fob)
    ctx._ck(ctx.lib.gcnx_gemm_dw2(ctx.h, _p(xa), xa.ld, _p(dha), dha.ld, _p(dwa), fia, foa, _p(xb), xb.ld, _p(dhb), dhb.ld,
                                  _p(dwb), fib, fob, n, L.PRECS[prec], _p(params), _p(grads),
                                  params.size if params is not None else (grads.size if grads is not None else 0), float(lr),
                                  C.byref(pending) if pending is not None else None, C.byref(leaf) if leaf is not None else None))


def pool_bwd_colsum(ctx, seg, dpooled, y, db, mode="sum"):
    """db = column sums of pool'(dpooled) * [y > 0] without materialising it (gcnx_pool_bwd_colsum)."""
    ctx._ck(ctx.lib.gcnx_pool_bwd_colsum(ctx.h, seg.dev.ptr, seg.n_graphs, _p(dpooled), dpooled.ld, _p(y), y.ld,
                                         y.shape[1], L.POOLS[mode], _p(db)))
    return db


def softmax_cce(ctx, logits, y, probs, loss_acc, dlogits=None, denom=None, cce="logits"):
    """cce = "logits": the softmax_cross_entropy_with_logits form Keras runs inside tf.function (gcn.py:328-335);
    "probs": the renormalise-and-clip form it runs on eager tensors (evaluate, gcn.py:351-354)."""
    b, c = logits.shape
    ctx._ck(ctx.lib.gcnx_softmax_cce(ctx.h, _p(logits), _p(y), b, c, float(denom if denom else b), _p(probs),
                                     _p(loss_acc), _p(dlogits), L.CCES[cce]))


def dense_softmax_cce(ctx, pooled, w, bias, y, probs, loss_acc=None, denom=None, dw=None, db=None, dpooled=None,
                      cce="logits"):
    """The classifier head in one launch (gcnx_dense_softmax_cce): probs = softmax(pooled W + b); with labels y
    also loss_acc = [CCE sum / denom, #correct] (overwritten); with dw also dw, db, dpooled."""
    b, h = pooled.shape
    c = w.shape[1]
    ctx._ck(ctx.lib.gcnx_dense_softmax_cce(ctx.h, _p(pooled), pooled.ld, _p(w), _p(bias), _p(y), b, h, c,
                                           float(denom if denom else max(b, 1)), _p(probs), _p(loss_acc), _p(dw), _p(db),
                                           _p(dpooled), dpooled.ld if dpooled is not None else 0, L.CCES[cce]))
    return probs


def act_bias_grad(ctx, dy, y, dz, act, db=None, alpha=None, dalpha=None):
    n, f = dy.shape
    ctx._ck(ctx.lib.gcnx_act_bias_grad(ctx.h, _p(dy), dy.ld, _p(y), y.ld if y is not None else 0, _p(dz), dz.ld, n,
                                       f, L.ACTS[act], _p(alpha), _p(db), _p(dalpha)))
    return dz


def gemm_dw(ctx, x, dh, dw, prec="f32"):
    n, fi = x.shape
    fo = dh.shape[1]
    assert dh.shape[0] == n and dw.shape == (fi, fo) and dw.contiguous
    ctx._ck(ctx.lib.gcnx_gemm_dw(ctx.h, _p(x), x.ld, _p(dh), dh.ld, _p(dw), n, fi, fo, L.PRECS[prec]))
    return dw


def gemm_dx(ctx, dh, w, dx, prec="f32", accumulate=False, y_mask=None, db=None, mask_bits=None):
    """dx = dh w^T (* [y_mask > 0], db = column sums).  mask_bits: the bit image gemm_relu_bits wrote for the same
    activation -- read instead of y_mask (gcnx_gemm_dx_bits)."""
    n, fo = dh.shape
    fi = w.shape[0]
    assert w.shape[1] == fo and dx.shape == (n, fi)
    if mask_bits is not None:
        assert not accumulate
        ctx._ck(ctx.lib.gcnx_gemm_dx_bits(ctx.h, _p(dh), dh.ld, _p(w), _p(dx), dx.ld, n, fi, fo, L.PRECS[prec], mask_bits.ptr, _p(db)))
        return dx
    ctx._ck(ctx.lib.gcnx_gemm_dx(ctx.h, _p(dh), dh.ld, _p(w), _p(dx), dx.ld, n, fi, fo, L.PRECS[prec],
                                 1 if accumulate else 0, _p(y_mask), y_mask.ld if y_mask is not None else 0, _p(db)))
    return dx


def gemm_dw_sgd(ctx, x, dh, dw, params, grads, lr, prec="f32", pending=None):
    """dw = X^T dH (a view into the flat ``grads``), then params -= lr * grads over the whole flat buffers
    (gcnx_gemm_dw_sgd: the split-K reduction launch applies the update, and finishes a ``pending`` reduction left
    by dense_bwd_deferred)."""
    n, fi = x.shape
    fo = dh.shape[1]
    assert dh.shape[0] == n and dw.shape == (fi, fo) and dw.contiguous and params.size == grads.size
    ctx._ck(ctx.lib.gcnx_gemm_dw_sgd(ctx.h, _p(x), x.ld, _p(dh), dh.ld, _p(dw), n, fi, fo, L.PRECS[prec], _p(params),
                                     _p(grads), params.size, float(lr), C.byref(pending) if pending is not None else None))


def dense_bwd_scratch_floats(ctx, n, fi, fo):
    return int(ctx.lib.gcnx_dense_bwd_scratch_floats(ctx.h, n, fi, fo))


def dense_bwd_deferred(ctx, x, dh, w, dx, dw, scratch, prec="f32", y_mask=None, db_prev=None):
    """dense_bwd whose reductions (db_prev, dw) are left in ``scratch`` for the step's last launch; returns the
    PendingReduce to hand to gemm_dw_sgd (all zeros if the fused form did not apply: results are then final)."""
    n, fo = dh.shape
    fi = w.shape[0]
    assert w.shape[1] == fo and dx.shape == (n, fi) and x.shape == (n, fi) and dw.shape == (fi, fo) and w.contiguous
    pend = L.PendingReduce()
    ctx._ck(ctx.lib.gcnx_dense_bwd_deferred(ctx.h, _p(x), x.ld, _p(dh), dh.ld, _p(w), n, fi, fo, L.PRECS[prec], _p(dx), dx.ld,
                                            _p(y_mask), y_mask.ld if y_mask is not None else 0, _p(db_prev), _p(dw),
                                            _p(scratch), scratch.size if scratch is not None else 0, C.byref(pend)))
    return pend


def dense_bwd(ctx, x, dh, w, dx, dw, prec="f32", y_mask=None, db_prev=None):
    """Backward of H = X W given dH in one call (gcnx_dense_bwd): dw = X^T dH, dx = dH W^T (* [y_mask > 0]),
    db_prev = column sums of dx."""
    n, fo = dh.shape
    fi = w.shape[0]
    assert w.shape[1] == fo and dx.shape == (n, fi) and x.shape == (n, fi) and dw.shape == (fi, fo) and w.contiguous
    ctx._ck(ctx.lib.gcnx_dense_bwd(ctx.h, _p(x), x.ld, _p(dh), dh.ld, _p(w), n, fi, fo, L.PRECS[prec], _p(dx), dx.ld,
                                   _p(y_mask), y_mask.ld if y_mask is not None else 0, _p(db_prev), _p(dw)))
    return dx


def sgd(ctx, params, grads, lr):
    ctx._ck(ctx.lib.gcnx_sgd(ctx.h, _p(params), _p(grads), params.size, float(lr)))


BN_EPS, BN_MOMENTUM = 1e-3, 0.99     # Keras BatchNormalization defaults


def bn_stats(ctx, z, sums, shift=None):
    n, f = z.shape
    ctx._ck(ctx.lib.gcnx_bn_stats(ctx.h, _p(z), z.ld, n, f, _p(shift), _p(sums)))


def bn_finalize(ctx, sums, count, mean, inv, moving_mean=None, moving_var=None, shift=None, momentum=BN_MOMENTUM, eps=BN_EPS):
    ctx._ck(ctx.lib.gcnx_bn_finalize(ctx.h, _p(sums), float(count), mean.size, momentum, eps, _p(shift), _p(mean), _p(inv),
                                     _p(moving_mean), _p(moving_var)))


def bn_moments(ctx, z, sums, mean, inv, moving_mean=None, moving_var=None, momentum=BN_MOMENTUM, eps=BN_EPS):
    """Batch mean / biased variance the way tf.nn.moments takes them (variance of the centred data): a first
    pass for the mean, a second one centred on it (gcnx_bn_moments; `sums` is kept for signature compatibility
    with the bn_stats / bn_finalize pair used for sync-BN)."""
    n, f = z.shape
    ctx._ck(ctx.lib.gcnx_bn_moments(ctx.h, _p(z), z.ld, n, f, momentum, eps, _p(mean), _p(inv), _p(moving_mean),
                                    _p(moving_var)))


def wimage_elems(ctx, fi, fo, transpose, prec):
    """bf16 elements of the weight image of W [fi, fo] (or of W^T) for gemm_wimage; 0 for prec "f32"."""
    return int(ctx.lib.gcnx_wimage_elems(int(fi), int(fo), 1 if transpose else 0, L.PRECS[prec]))


def wimage_prepare(ctx, jobs):
    """ONE launch writes the images of several matrices: jobs = [(w, img, transpose, prec), ...] (gcnx_wimage_prepare)."""
    if not jobs:
        return
    arr = (L.WimageJob * len(jobs))()
    for k, (w, img, transpose, prec) in enumerate(jobs):
        fi, fo = w.shape
        arr[k] = L.WimageJob(w.ptr, img.ptr, fi, fo, 1 if transpose else 0, L.PRECS[prec])
    ctx._ck(ctx.lib.gcnx_wimage_prepare(ctx.h, len(jobs), C.cast(arr, C.c_void_p)))


def gemm_wimage(ctx, x, img, fi, fo, out, transpose=False, bias=None, prec="bf16x3", accumulate=False, bn_parts=None):
    """out = x W + bias (transpose False) / out (+)= x W^T (True) with W read from its image (gcnx_gemm_wimage).  Returns
    None when the kernel does not serve the shape (nothing launched), else the number of batch-norm parts written to
    ``bn_parts`` (0 without)."""
    n = x.shape[0]
    nparts = C.c_int32(0)
    rc = ctx.lib.gcnx_gemm_wimage(ctx.h, _p(x), x.ld, img.ptr, int(fi), int(fo), 1 if transpose else 0, _p(bias), _p(out), out.ld, n,
                                  L.PRECS[prec], 1 if accumulate else 0, _p(bn_parts), C.byref(nparts) if bn_parts is not None else None)
    if rc == L.ERR_UNSUPPORTED:
        return None
    ctx._ck(rc)
    return nparts.value


def gemm_wimage_parts(ctx, n):
    return int(ctx.lib.gcnx_gemm_wimage_parts(ctx.h, int(n)))


def bn_finalize_parts(ctx, parts, nparts, mean, inv, moving_mean=None, moving_var=None, momentum=BN_MOMENTUM, eps=BN_EPS):
    """mean / inv (and the moving statistics) from the (rows, mean, M2) parts gemm_wimage left (gcnx_bn_finalize_parts)."""
    ctx._ck(ctx.lib.gcnx_bn_finalize_parts(ctx.h, _p(parts), int(nparts), mean.size, momentum, eps, _p(mean), _p(inv),
                                           _p(moving_mean), _p(moving_var)))


def bn_act(ctx, z, mean, inv, gamma, beta, y, act=None, alpha=None):
    n, f = z.shape
    ctx._ck(ctx.lib.gcnx_bn_act(ctx.h, _p(z), z.ld, n, f, _p(mean), _p(inv), _p(gamma), _p(beta), L.ACTS[act], _p(alpha),
                                _p(y), y.ld))
    return y


def bn_act_bwd(ctx, dy, z, mean, inv, gamma, beta, dz, scratch, act=None, alpha=None, training=True, dgamma=None, dbeta=None,
               dalpha=None):
    n, f = z.shape
    ctx._ck(ctx.lib.gcnx_bn_act_bwd(ctx.h, _p(dy), dy.ld, _p(z), z.ld, n, f, _p(mean), _p(inv), _p(gamma), _p(beta),
                                    L.ACTS[act], _p(alpha), 1 if training else 0, _p(dz), dz.ld, _p(dgamma), _p(dbeta),
                                    _p(dalpha), _p(scratch)))
    return dz


def bn_act_bwd_stats(ctx, dy, z, mean, inv, gamma, beta, scratch, act=None, alpha=None, dgamma=None, dbeta=None, dalpha=None):
    """First half of bn_act_bwd (sync-BN): the three local column sums -> scratch[3f] and the parameter gradients."""
    n, f = z.shape
    ctx._ck(ctx.lib.gcnx_bn_act_bwd_stats(ctx.h, _p(dy), dy.ld, _p(z), z.ld, n, f, _p(mean), _p(inv), _p(gamma), _p(beta),
                                          L.ACTS[act], _p(alpha), _p(scratch), _p(dgamma), _p(dbeta), _p(dalpha)))


def bn_act_bwd_apply(ctx, dy, z, mean, inv, gamma, beta, sums, count, dz, act=None, alpha=None, training=True):
    """Second half: dz from the (all-reduced) sums and the global row count."""
    n, f = z.shape
    ctx._ck(ctx.lib.gcnx_bn_act_bwd_apply(ctx.h, _p(dy), dy.ld, _p(z), z.ld, n, f, _p(mean), _p(inv), _p(gamma), _p(beta),
                                          L.ACTS[act], _p(alpha), _p(sums), float(count), 1 if training else 0, _p(dz), dz.ld))
    return dz


def dropout(ctx, x, rate, seed, stream_id, step=None, out=None):
    """Keras Dropout(rate), training mode (gcnx_dropout): out = x * keep / (1 - rate); the same call on the incoming gradient
    is the backward pass (the mask is a stateless hash of (seed, stream_id, step, element)).  In place by default."""
    out = x if out is None else out
    n, f = x.shape
    assert out.shape == (n, f)
    ctx._ck(ctx.lib.gcnx_dropout(ctx.h, _p(x), x.ld, n, f, float(rate), int(seed) & 0xFFFFFFFF, int(stream_id) & 0xFFFFFFFF, _p(step),
                                 _p(out), out.ld))
    return out


def counter_add(ctx, counter, inc=1):
    """*counter += inc on the stream (a uint32 device scalar: the optimizer's step count)."""
    ctx._ck(ctx.lib.gcnx_counter_add(ctx.h, _p(counter), int(inc)))


def add(ctx, a, b, out):
    """out = a + b (GeneralGNN(connectivity="sum"))."""
    n, f = a.shape
    assert b.shape == (n, f) and out.shape == (n, f)
    ctx._ck(ctx.lib.gcnx_add(ctx.h, _p(a), a.ld, _p(b), b.ld, _p(out), out.ld, n, f))
    return out


def spmm_minmax(ctx, a, h, out, cnt=None, mode="max"):
    """GeneralConv(aggregate="max" | "min" | "prod") (gcnx_spmm_csr_minmax / gcnx_spmm_csr_prod): out[t] = max / min / product over
    the entries (t, s) of h[s] (values ignored); cnt = what the gradient needs -- the number of entries attaining the extremum, or
    for "prod" the product of the row's non-zero messages (0 where two or more are zero)."""
    n, f = h.shape
    assert a.n == n and out.shape == (n, f) and mode in ("max", "min", "prod")
    if mode == "prod":
        ctx._ck(ctx.lib.gcnx_spmm_csr_prod(ctx.h, a.rowptr.ptr, a.colidx.ptr, _p(h), h.ld, _p(out), out.ld, _p(cnt),
                                           cnt.ld if cnt is not None else 0, n, f))
        return out
    ctx._ck(ctx.lib.gcnx_spmm_csr_minmax(ctx.h, a.rowptr.ptr, a.colidx.ptr, _p(h), h.ld, _p(out), out.ld, _p(cnt),
                                         cnt.ld if cnt is not None else 0, n, f, 1 if mode == "min" else 0))
    return out


def spmm_minmax_bwd(ctx, at, h, out, cnt, dy, dh, mode="max"):
    """Gradient of spmm_minmax wrt h (gcnx_spmm_csr_minmax_bwd / gcnx_spmm_csr_prod_bwd); ``at`` is the transposed operator."""
    n, f = h.shape
    assert at.n == n and out.shape == (n, f) and cnt.shape == (n, f) and dy.shape == (n, f) and dh.shape == (n, f)
    if mode == "prod":
        ctx._ck(ctx.lib.gcnx_spmm_csr_prod_bwd(ctx.h, at.rowptr.ptr, at.colidx.ptr, _p(h), h.ld, _p(out), out.ld, _p(cnt), cnt.ld,
                                               _p(dy), dy.ld, _p(dh), dh.ld, n, f))
        return dh
    ctx._ck(ctx.lib.gcnx_spmm_csr_minmax_bwd(ctx.h, at.rowptr.ptr, at.colidx.ptr, _p(h), h.ld, _p(out), out.ld, _p(cnt), cnt.ld,
                                             _p(dy), dy.ld, _p(dh), dh.ld, n, f))
    return dh
