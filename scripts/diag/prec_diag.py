#!/usr/bin/env python3
"""Which product carries the bf16x3 error of the config-3 step?  One full-size step per variant, each
with ONE of the five weight GEMMs (or a group) forced to another precision; gradients against the fp32 C oracle."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import gcnx
from gcnx import device as D, synth, models
from gcnx.device import DeviceCSR, Segments
from gcnx.models import DeviceBatch, GCN2
from oracle import c_oracle

ORDER = ("w1", "b1", "w2", "b2", "w3", "b3")
hb = synth.block_diag_batch()
hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
ctx = gcnx.Context(0)
a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
m = GCN2(ctx, 2, hidden=256, seed=0, use_graph=False)
m.build(hb.f)
w0 = m.get_weights()
flat0 = np.concatenate([w.ravel() for w in w0])
cpu = c_oracle.Gcn2Cpu(hb, 256, 2, flat0)
rl, ra = cpu.step(lr=0.0)
rg = cpu.grads.copy()

force = {}
count = {}
orig = {k: getattr(D, k) for k in ("gemm", "gemm_relu_bits", "gemm_dw", "gemm_dx")}
def wrap(name, site_of):
    def f(*args, **kw):
        count[name] = count.get(name, 0) + 1
        site = site_of(count[name])
        if site in force:
            kw["prec"] = force[site]
            if name == "gemm_relu_bits" and force[site] == "f32":
                return False                      # the model then calls gemm(act=relu): counted as site fwd1 there
            if name == "gemm_dx" and force[site] == "f32":
                kw["mask_bits"] = None
        return orig[name](*args, **kw)
    return f
# call order in a step: gemm_relu_bits (fwd1) [or gemm #1 when refused], gemm (fwd2), gemm_dw #1 (dW2), gemm_dx (dX), gemm_dw #2 (dW1)
state = {"bits_refused": False}
def gemm_site(i):
    return "fwd1" if (state["bits_refused"] and i == 1) else "fwd2"
D.gemm = wrap("gemm", gemm_site)
D.gemm_relu_bits = wrap("gemm_relu_bits", lambda i: "fwd1")
D.gemm_dw = wrap("gemm_dw", lambda i: "dW2" if i == 1 else "dW1")
D.gemm_dx = wrap("gemm_dx", lambda i: "dX")

def rel(x, r):
    return float(np.max(np.abs(x.astype(np.float64) - r)) / max(float(np.max(np.abs(r))), 1e-30))

rows = []
variants = [("all bf16x3", {}), ("fwd1 f32", {"fwd1": "f32"}), ("fwd2 f32", {"fwd2": "f32"}), ("dW2 f32", {"dW2": "f32"}),
            ("dX f32", {"dX": "f32"}), ("dW1 f32", {"dW1": "f32"}), ("fwd f32", {"fwd1": "f32", "fwd2": "f32"}),
            ("bwd f32", {"dW2": "f32", "dX": "f32", "dW1": "f32"}), ("dW1+dW2 f32", {"dW1": "f32", "dW2": "f32"}),
            ("all f32", {"fwd1": "f32", "fwd2": "f32", "dW2": "f32", "dX": "f32", "dW1": "f32"})]
for name, fz in variants:
    force.clear(); force.update(fz); count.clear()
    state["bits_refused"] = fz.get("fwd1") == "f32"
    m.prec = "bf16x3"; m._drop_graphs(); m.set_weights(w0)
    loss, acc = m.train_step(batch, None, lr=0.0)
    got = np.concatenate([m.gradients()[k].ravel() for k in ORDER])
    off = 0; errs = {}
    for k, w in zip(ORDER, w0):
        errs[k] = rel(got[off:off + w.size], rg[off:off + w.size]); off += w.size
    rows.append((name, abs(loss - rl), errs))
    print(f"{name:14s} dloss {abs(loss-rl):.2e}  " + "  ".join(f"{k} {v:.2e}" for k, v in errs.items()), flush=True)
ctx.close()
