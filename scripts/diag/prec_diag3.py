#!/usr/bin/env python3
"""Config-3 step against an fp64 reference with the ReLU kinks separated: reference gradients evaluated (a) with the
reference's own masks, (b) with the DEVICE's masks [Y > 0] -- and, for plain bf16, an fp64 model whose GEMM operands are
rounded to bf16 in the REFERENCE's order A (X W)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import scipy.sparse as sp
import gcnx
from gcnx import synth
from gcnx.device import DeviceCSR, Segments
from gcnx.models import DeviceBatch, GCN2
from oracle import gcn_oracle as O

ORDER = ("w1", "b1", "w2", "b2", "w3", "b3")
hb = synth.block_diag_batch()
hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
ctx = gcnx.Context(0)
a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
m = GCN2(ctx, 2, hidden=256, seed=0, use_graph=False)
m.build(hb.f)
w0 = m.get_weights()
A = sp.csr_matrix((hb.vals.astype(np.float64), hb.colidx, hb.rowptr), shape=(hb.n, hb.n))
AT = A.T.tocsr()
P = sp.csr_matrix((np.ones(hb.n), (np.repeat(np.arange(hb.n_graphs), np.diff(hb.graph_ptr)), np.arange(hb.n))), shape=(hb.n_graphs, hb.n))
p = {k: w.astype(np.float64) for k, w in zip(ORDER, w0)}
x = hb.x.astype(np.float64); y = hb.y.astype(np.float64); B = hb.n_graphs

def rb(v):     # round to bf16 (RNE), back to fp64
    return O.bf16_from_bits(O.bf16_bits(v.astype(np.float32))).astype(np.float64)

def forward(r=lambda v: v):
    z1 = A @ (r(x) @ r(p["w1"])) + p["b1"]; y1 = np.maximum(z1, 0)
    z2 = A @ (r(y1) @ r(p["w2"])) + p["b2"]; y2 = np.maximum(z2, 0)
    pooled = P @ y2
    logits = pooled @ p["w3"] + p["b3"]
    return z1, y1, z2, y2, pooled, logits

def backward(fw, m1, m2, r=lambda v: v):
    z1, y1, z2, y2, pooled, logits = fw
    zz = logits - logits.max(1, keepdims=True); pr = np.exp(zz); pr /= pr.sum(1, keepdims=True)
    loss = float((np.log(np.exp(zz).sum(1)) - (y * zz).sum(1)).sum() / B)
    dl = (pr - y) / B
    g = {"w3": pooled.T @ dl, "b3": dl.sum(0)}
    dz2 = (P.T @ (dl @ p["w3"].T)) * m2
    g["b2"] = dz2.sum(0); dh2 = AT @ dz2
    g["w2"] = r(y1).T @ r(dh2)
    dz1 = (r(dh2) @ r(p["w2"]).T) * m1
    g["b1"] = dz1.sum(0); dh1 = AT @ dz1
    g["w1"] = r(x).T @ r(dh1)
    return loss, np.concatenate([g[k].ravel() for k in ORDER])

def rel(v, r):
    return float(np.max(np.abs(v.astype(np.float64) - r)) / max(float(np.max(np.abs(r))), 1e-30))
def report(name, got, ref):
    off = 0; out = []
    for k, w in zip(ORDER, w0):
        out.append(f"{k} {rel(got[off:off + w.size], ref[off:off + w.size]):.2e}"); off += w.size
    print(f"{name:44s} " + "  ".join(out), flush=True)

t0 = time.time()
fw = forward()
loss64, ref_own = backward(fw, fw[1] > 0, fw[3] > 0)
fwb = forward(rb)
print(f"fp64 forward + backward {time.time()-t0:.1f} s, loss {loss64:.6f}", flush=True)
for prec in ("f32", "bf16x3", "bf16"):
    m.prec = prec; m._drop_graphs(); m.set_weights(w0)
    loss, acc = m.train_step(batch, None, lr=0.0)
    got = np.concatenate([m.gradients()[k].ravel() for k in ORDER])
    bufs = m._bufs
    m1 = bufs["y1"].numpy() > 0; m2 = bufs["y2"].numpy() > 0
    f1 = int((m1 != (fw[1] > 0)).sum()); f2 = int((m2 != (fw[3] > 0)).sum())
    zmax1 = float(np.abs(fw[0][m1 != (fw[1] > 0)]).max()) if f1 else 0.0
    zmax2 = float(np.abs(fw[2][m2 != (fw[3] > 0)]).max()) if f2 else 0.0
    print(f"{prec}: loss err {abs(loss-loss64):.2e}; mask flips layer1 {f1} (|z| <= {zmax1:.2e}, rms z {np.sqrt((fw[0]**2).mean()):.2e}), layer2 {f2} (|z| <= {zmax2:.2e}, rms {np.sqrt((fw[2]**2).mean()):.2e})", flush=True)
    report(f"  {prec} vs fp64, reference masks", got, ref_own)
    _, ref_dev = backward(fw, m1, m2)
    report(f"  {prec} vs fp64, device masks", got, ref_dev)
    if prec == "bf16":
        _, ref_b = backward(fwb, m1, m2, rb)
        report("  bf16 vs fp64 bf16-operand model (ref order), device masks", got, ref_b)
        _, ref_b2 = backward(fwb, fwb[1] > 0, fwb[3] > 0, rb)
        report("  bf16 vs fp64 bf16-operand model (ref order), own masks", got, ref_b2)
        report("  bf16-operand model vs exact (own masks each)", ref_b2, ref_own)
ctx.close()
