#!/usr/bin/env python3
"""Config-3 step: device (f32 / bf16x3 / bf16) and the fp32 C oracle, each against an fp64 reference (the NumPy oracle's
gcn2_loss_and_grads with scipy.sparse doing the aggregation)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import scipy.sparse as sp
import gcnx
from gcnx import synth
from gcnx.device import DeviceCSR, Segments
from gcnx.models import DeviceBatch, GCN2
from oracle import c_oracle, gcn_oracle as O

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

# fp64 reference
t0 = time.time()
A = sp.csr_matrix((hb.vals.astype(np.float64), hb.colidx, hb.rowptr), shape=(hb.n, hb.n))
AT = A.T.tocsr()
O.spmm_csr = lambda rp, ci, v, h: A @ h
O.spmm_csr_T = lambda rp, ci, v, g: AT @ g
p64 = {k: w.astype(np.float64) for k, w in zip(ORDER, w0)}
loss64, acc64, g64, cache = O.gcn2_loss_and_grads(p64, hb.x.astype(np.float64), (hb.rowptr, hb.colidx, hb.vals), hb.graph_ptr, hb.y)
ref = np.concatenate([g64[k].ravel() for k in ORDER])
print(f"fp64 reference: loss {loss64:.9f} acc {acc64:.5f} in {time.time()-t0:.1f} s; |logits| max {np.abs(cache['logits']).max():.2f}", flush=True)

def rel(x, r):
    return float(np.max(np.abs(x.astype(np.float64) - r)) / max(float(np.max(np.abs(r))), 1e-30))
def report(name, loss, got):
    off = 0; out = []
    for k, w in zip(ORDER, w0):
        out.append(f"{k} {rel(got[off:off + w.size], ref[off:off + w.size]):.2e}"); off += w.size
    print(f"{name:28s} dloss {abs(loss - loss64):.2e} ({abs(loss-loss64)/abs(loss64):.1e} rel)  " + "  ".join(out), flush=True)

cpu = c_oracle.Gcn2Cpu(hb, 256, 2, flat0)
rl, ra = cpu.step(lr=0.0); report("C oracle fp32", rl, cpu.grads.copy())
cpu.params[:] = flat0
rl, ra = cpu.step(lr=0.0, bf16_operands=True, layer1_s_order=True); report("C oracle bf16 model (S order)", rl, cpu.grads.copy())
cpu.params[:] = flat0
rl, ra = cpu.step(lr=0.0, bf16_operands=True, layer1_s_order=False); report("C oracle bf16 model (ref order)", rl, cpu.grads.copy())
for prec in ("f32", "bf16x3", "bf16"):
    m.prec = prec; m._drop_graphs(); m.set_weights(w0)
    loss, acc = m.train_step(batch, None, lr=0.0)
    report("device " + prec, loss, np.concatenate([m.gradients()[k].ravel() for k in ORDER]))
os.environ["GCNX_S_ORDER"] = "0"
m2 = GCN2(ctx, 2, hidden=256, seed=0, use_graph=False); m2.build(hb.f)
for prec in ("f32", "bf16x3", "bf16"):
    m2.prec = prec; m2._drop_graphs(); m2.set_weights(w0)
    loss, acc = m2.train_step(batch, None, lr=0.0)
    report("device " + prec + " (ref order)", loss, np.concatenate([m2.gradients()[k].ravel() for k in ORDER]))
ctx.close()
