#!/usr/bin/env python3
"""Full-size (config 3) comparison of the bf16-storage path with fp32 storage, tensor by tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd")); sys.path.insert(0, ROOT)
import numpy as np
import gcnx
from gcnx import device as D, synth
from gcnx.device import DeviceCSR, Segments
from gcnx.models import DeviceBatch, GCN2
from oracle import gcn_oracle as O

ctx = gcnx.Context(0)
sizes, pairs = synth.block_diag_plan()
hb = synth.block_diag_shard(0, len(sizes), sizes, pairs, 256, seed=2)
vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
keep = {}
for store16 in (True, False):
    m = GCN2(ctx, 2, hidden=256, seed=0, prec="bf16", use_graph=False)
    m._knob["act16"] = store16
    loss, acc = m.train_step(batch, None, lr=0.0)
    print("try", m._act16_try(batch, "grads"), {k: m._knob[k] for k in ("act16", "side", "s_order")}, m.prec, m.hidden, m.f_in, batch.n,
          batch.a.plan is not None, batch.a.vals is not None, m._bufs.get("act16"), flush=True)
    if store16 and not m._bufs.get("act16"):
        t16 = ctx.empty((hb.n, 256), np.uint16)
        print("direct spmm_bf16out:", D.spmm_bf16out(ctx, batch.a, batch.x, None, t16), flush=True)
        sys.exit(1)
    b = m._bufs
    t = {"loss": loss, "y2": b["y2"].numpy(), "y1bits": b["y1bits"].numpy().copy(), "y2bits": b["y2bits"].numpy().copy()}
    if store16:
        t["s1"] = b["s1_16"].numpy(); t["y1"] = b["y1_16"].numpy()
        t["dh2"] = m._cap.view("dh2_16", hb.n, 256, np.uint16).numpy(); t["dz1"] = m._cap.view("dz1_16", hb.n, 256, np.uint16).numpy()
    else:
        t["s1"] = O.bf16_bits(b["s1"].numpy()); t["y1"] = O.bf16_bits(b["y1"].numpy())
        t["dh2"] = O.bf16_bits(b["h"].numpy()); t["dz1"] = O.bf16_bits(b["dz2"].numpy())
    t["g"] = {k: v.copy() for k, v in m.gradients().items()}
    keep[store16] = t
    print("store16", store16, "loss", loss, "act16", b.get("act16"), flush=True)
A, B = keep[True], keep[False]
for k in ("s1", "y1", "y1bits", "y2", "y2bits", "dh2", "dz1"):
    d = A[k] != B[k]
    print(f"{k:8s} differing elements {int(d.sum())} of {d.size}", end="")
    if d.any():
        idx = np.argwhere(d)
        print("  first", idx[0], "rows with differences", len(np.unique(idx[:, 0])), "row range", idx[:, 0].min(), idx[:, 0].max(), end="")
    print(flush=True)
for k in A["g"]:
    print("grad", k, "equal", np.array_equal(A["g"][k], B["g"][k]), float(np.abs(A["g"][k] - B["g"][k]).max()))
