#!/usr/bin/env python3
"""GEMM microbenchmark (tuning aid): the three GCNConv/Dense GEMM entry points over a list of shapes.
    python scripts/gemm_bench.py [--n 20498] [--prec f32] [--iters 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import gcnx
from gcnx import device as D
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20498)
ap.add_argument("--prec", default="f32")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--shapes", default="16x256,128x128,256x256,512x256,768x256,1024x256")
args = ap.parse_args()
ctx = gcnx.Context(0)
rng = np.random.default_rng(0)
n = args.n
for sh in args.shapes.split(","):
    fi, fo = map(int, sh.split("x"))
    x = ctx.to_device(rng.standard_normal((n, fi), dtype=np.float32))
    w = ctx.to_device((rng.standard_normal((fi, fo)) / np.sqrt(fi)).astype(np.float32))
    dh = ctx.to_device(rng.standard_normal((n, fo), dtype=np.float32))
    out, dx, dw, dbv = ctx.empty((n, fo)), ctx.empty((n, fi)), ctx.empty((fi, fo)), ctx.empty(fi)
    res = []
    for name, fn in (("fwd", lambda: D.gemm(ctx, x, w, None, out, prec=args.prec)),
                     ("dx", lambda: D.gemm_dx(ctx, dh, w, dx, prec=args.prec)),
                     ("dx+mask", lambda: D.gemm_dx(ctx, dh, w, dx, prec=args.prec, y_mask=x)),
                     ("dx+mask+db", lambda: D.gemm_dx(ctx, dh, w, dx, prec=args.prec, y_mask=x, db=dbv)),
                     ("dw", lambda: D.gemm_dw(ctx, x, dh, dw, prec=args.prec))):
        for _ in range(3): fn()
        e0 = ctx.event().record()
        for _ in range(args.iters): fn()
        e1 = ctx.event().record()
        us = e1.elapsed_ms_since(e0) / args.iters * 1e3
        res.append(f"{name} {us:7.1f} us {2.0*n*fi*fo/us/1e6:6.1f} TF/s")
    print(f"N={n} {fi:4d}x{fo:<4d} {args.prec}: " + "   ".join(res), flush=True)
ctx.close()
