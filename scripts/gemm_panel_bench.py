#!/usr/bin/env python3
"""Split-bf16 panel GEMMs (csrc/gemm_panel.hip) at GeneralGNN's shapes: us per call and bytes per second.
    python scripts/gemm_panel_bench.py [--n 22576] [--prec bf16x3] [--iters 30]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import gcnx
from gcnx import device as D
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=22576)
ap.add_argument("--prec", default="bf16x3")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--only", default="", help="fwd: only the X W product (profiling)")
ap.add_argument("--ks", default="256,512,768,1024")
args = ap.parse_args()
ctx = gcnx.Context(0)
rng = np.random.default_rng(0)
n = args.n
def timeit(fn):
    for _ in range(3): fn()
    e0 = ctx.event().record()
    for _ in range(args.iters): fn()
    return ctx.event().record().elapsed_ms_since(e0) / args.iters * 1e3
cat = ctx.to_device(rng.standard_normal((n, 1280), dtype=np.float32))
dz = ctx.to_device(rng.standard_normal((n, 256), dtype=np.float32))
out = ctx.empty((n, 256)); dcat = ctx.zeros((n, 1280))
parts = ctx.empty(3 * D.gemm_wimage_parts(ctx, n) * 256)
for k in [int(v) for v in args.ks.split(',')]:
    w = ctx.to_device((rng.standard_normal((k, 256)) / 16).astype(np.float32))
    img = ctx.empty(D.wimage_elems(ctx, k, 256, False, args.prec), np.uint16)
    imgt = ctx.empty(D.wimage_elems(ctx, k, 256, True, args.prec), np.uint16)
    D.wimage_prepare(ctx, [(w, img, False, args.prec), (w, imgt, True, args.prec)])
    x = cat.cols(0, k)
    ref = x.numpy().astype(np.float64) @ w.numpy().astype(np.float64)
    D.gemm_wimage(ctx, x, img, k, 256, out, prec=args.prec, bn_parts=parts)
    err = np.abs(out.numpy() - ref).max() / np.abs(ref).max()
    t_f = timeit(lambda: D.gemm_wimage(ctx, x, img, k, 256, out, prec=args.prec, bn_parts=parts))
    if args.only == "fwd":
        print(f"K={k}: X W {t_f:.1f} us"); continue
    t_x = timeit(lambda: D.gemm_wimage(ctx, dz, imgt, k, 256, dcat.cols(0, k), transpose=True, prec=args.prec, accumulate=True))
    dw = ctx.empty((k, 256))
    t_w = timeit(lambda: D.gemm_dw(ctx, x, dz, dw, prec=args.prec))
    refw = x.numpy().astype(np.float64).T @ dz.numpy().astype(np.float64)
    errw = np.abs(dw.numpy() - refw).max() / np.abs(refw).max()
    bf = 4 * n * (k + 256); bx = 4 * n * (256 + 2 * k)
    print(f"K={k:5d}: X W {t_f:7.1f} us ({bf/t_f/1e6:5.2f} TB/s, err {err:.1e})   dH W^T (+=) {t_x:7.1f} us ({bx/t_x/1e6:5.2f} TB/s)   "
          f"X^T dH {t_w:7.1f} us ({bf/t_w/1e6:5.2f} TB/s, err {errw:.1e})", flush=True)
ctx.close()
