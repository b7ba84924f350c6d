"""ctypes access to the C restatement (oracle/gcn_oracle.c).  TEST INFRASTRUCTURE ONLY --
PARITY UNPINNED (see the header of gcn_oracle.c).  Importers: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libgcn_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "gcn_oracle.c")
        if not os.path.exists(LIB) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB)):
            build()                                        # (make: a no-op when the library is current)
        _lib = C.CDLL(LIB)
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _f(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def threads():
    return load().orc_max_threads()


def spmm_csr(rowptr, colidx, vals, h, bias=None, relu=False):
    lib = load()
    n, f = h.shape
    out = np.empty((n, f), np.float32)
    lib.orc_spmm_csr(_f(rowptr), _f(colidx), _f(vals), _f(h), C.c_int64(f), _f(bias), _f(out), C.c_int64(f),
                     C.c_int32(n), C.c_int32(f), C.c_int(int(relu)))
    return out


def gemm(x, w, bias=None, relu=False):
    lib = load()
    n, fi = x.shape
    fo = w.shape[1]
    out = np.empty((n, fo), np.float32)
    lib.orc_gemm(_f(x), C.c_int64(fi), _f(w), _f(bias), _f(out), C.c_int64(fo), C.c_int64(n), C.c_int32(fi),
                 C.c_int32(fo), C.c_int(int(relu)))
    return out


class Gcn2Cpu:
    """The M1 model step on the CPU (fp32), for parity checks and the cpu_baseline timing."""

    def __init__(self, batch, hidden, n_classes, params_flat):
        self.lib = load()
        self.b = batch
        self.h, self.c = hidden, n_classes
        self.params = np.ascontiguousarray(params_flat, np.float32).copy()
        self.grads = np.zeros_like(self.params)
        n, bb = batch.n, batch.n_graphs
        self.work = np.empty(4 * n * hidden + 2 * bb * hidden + 3 * bb * n_classes, np.float32)
        self.out = np.zeros(2, np.float32)
        self.rowptr = np.ascontiguousarray(batch.rowptr, np.int32)
        self.colidx = np.ascontiguousarray(batch.colidx, np.int32)
        self.vals = None if batch.vals is None else np.ascontiguousarray(batch.vals, np.float32)
        self.gp = np.ascontiguousarray(batch.graph_ptr, np.int32)
        self.x = np.ascontiguousarray(batch.x, np.float32)
        self.y = np.ascontiguousarray(batch.y, np.float32)

    def step(self, lr=0.0, denom=None, cce="logits", bf16_operands=False, layer1_s_order=False):
        """cce "logits": the loss train_step computes under tf.function; "probs": the eager clip form.
        bf16_operands: every dense product sees bf16-rounded operands (fp32 accumulate) -- the GCNX_PREC_BF16 model.
        layer1_s_order: layer 1 as (A X) W1 -- the device's order when F <= H; matters for the bf16-operand model only
        (which operands get rounded), the fp32 checks run against the reference's order A (X W1)."""
        b = self.b
        self.lib.orc_set_bf16_operands(C.c_int(1 if bf16_operands else 0))
        s1 = np.empty(b.n * b.f, np.float32) if layer1_s_order else None
        self.lib.orc_set_layer1_s_order(_f(s1))
        self.lib.orc_gcn2_step(_f(self.rowptr), _f(self.colidx), _f(self.vals), _f(self.gp), _f(self.x), _f(self.y),
                               C.c_int32(b.n), C.c_int32(b.n_graphs), C.c_int32(b.f), C.c_int32(self.h),
                               C.c_int32(self.c), _f(self.params), _f(self.grads), C.c_float(lr),
                               C.c_float(denom or b.n_graphs), _f(self.work), _f(self.out),
                               C.c_int(1 if cce == "logits" else 0))
        self.lib.orc_set_bf16_operands(C.c_int(0))
        self.lib.orc_set_layer1_s_order(None)
        return float(self.out[0]), float(self.out[1]) / b.n_graphs
