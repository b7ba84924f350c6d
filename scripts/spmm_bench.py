#!/usr/bin/env python3
"""SpMM-only microbenchmark (tuning aid): one process, interleaved variants, HIP-event timing.
    python scripts/spmm_bench.py [--workload block1m|ecoli|powerlaw] [--iters 20] [--slabs 256,128,64]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import gcnx
from gcnx import device as D, synth
from gcnx.device import DeviceCSR

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="block1m")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--slabs", default="0")
ap.add_argument("--unweighted", action="store_true")
ap.add_argument("--shard", default="", help="r,w: rank r's shard of w (block1m; the cut bench.py makes)")
ap.add_argument("--pads", default="", help="comma list of MiB: re-time with h and out re-allocated that far apart (HBM channel phase of the two streams)")
ap.add_argument("--hog", type=int, default=0, help="GiB allocated (and kept) before anything else: does where the arrays land matter?")
ap.add_argument("--hog-after-csr", type=int, default=0)
ap.add_argument("--cb", type=int, default=-1, help="spmm_cb knob for the whole process (tuning build: 8 + mask = only those kinds of column-block items)")
ap.add_argument("--tile-wgs", type=int, default=0, help="workgroups of the 1024-thread tile launch (spmm_tile_wgs; 0: one per CU)")
ap.add_argument("--conc", type=int, default=-1, help="1 / 0: the plan path's launches as concurrent branches or not (spmm_conc)")
args = ap.parse_args()
if args.workload == "block1m":       # the batches bench.py times (per-graph-seeded generators)
    sizes, pairs = synth.block_diag_plan()
    lo, hi = 0, len(sizes)
    if args.shard:
        from gcnx import shard as _sh
        r, w = (int(v) for v in args.shard.split(","))
        bounds = _sh.partition_by_cost(sizes + 2 * pairs + sizes, w)
        lo, hi = int(bounds[r]), int(bounds[r + 1])
    hb = synth.block_diag_shard(lo, hi, sizes, pairs, 256, seed=2, with_x=False); f = 256
elif args.workload == "ecoli":
    hb = synth.ecoli_shard(0, 32, 128, seed=1); f = 128
else:
    hb = synth.power_law_batch(122, 8192, 256, seed=3, with_x=False, first_graph=0); f = 256
vals = None if args.unweighted else synth.gcn_norm_host(hb.rowptr, hb.colidx)
ctx = gcnx.Context(0)
_hog = [ctx.empty((1 << 28,)) for _ in range(args.hog)]
if args.conc >= 0: ctx.set_tuning("spmm_conc", args.conc)
if args.tile_wgs > 0: ctx.set_tuning("spmm_tile_wgs", args.tile_wgs)
a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
_hog2 = [ctx.empty((1 << 28,)) for _ in range(args.hog_after_csr)]
h = ctx.to_device(np.random.default_rng(0).standard_normal((hb.n, f), dtype=np.float32))
out = ctx.empty((hb.n, f)); bias = ctx.zeros(f)
alg = synth.spmm_algorithmic_bytes(hb.n, hb.nnz, f, vals is not None)
print(f"N={hb.n} nnz={hb.nnz} F={f} alg={alg/1e6:.1f} MB")
h16 = D.to_bf16(ctx, h) if f in (64, 128, 256) else None
o16 = ctx.empty((hb.n, f), np.uint16) if h16 is not None else None
alg16 = 4 * (hb.n + 1) + (8 if vals is not None else 4) * hb.nnz + 4 * hb.n * f
for rnd in range(args.rounds):
    for slab in args.slabs.split(","):
        if slab == "bf16":               # the bf16-feature kernel (gcnx_spmm_csr_bf16), on its own algorithmic bytes
            for _ in range(3): D.spmm_bf16(ctx, a, h16, bias, o16, act="relu")
            e0 = ctx.event().record()
            for _ in range(args.iters): D.spmm_bf16(ctx, a, h16, bias, o16, act="relu")
            ms = ctx.event().record().elapsed_ms_since(e0) / args.iters
            print(f"round {rnd} slab     bf16: {ms*1e3:9.1f} us  {alg16/ms/1e6:8.1f} GB/s  frac {alg16/ms/1e6/8000:.3f}", flush=True)
            continue
        # variants: "0" = what the library picks; "rows" / "tile" / "pipe" = that kernel; a number = rows kernel, that slab width
        if "GCNX_SPMM_KERNEL" not in os.environ:
            ctx.set_tuning("spmm_slab", 0)
            ctx.set_tuning("spmm_cb", 0 if slab == "cb0" else (args.cb if args.cb >= 0 else 1))      # "cb0": graphs of >= 4096 rows on the row gather + hub segments (r3)
            if slab == "cb0": ctx.set_tuning("spmm_kernel", "auto")
            elif slab in ("rows", "tile", "pipe"): ctx.set_tuning("spmm_kernel", slab)
            elif slab[0] == "t": ctx.set_tuning("spmm_slab", int(slab[1:])); ctx.set_tuning("spmm_kernel", "auto")   # tiers, tall graphs' row chunks in slabs
            elif slab != "0": ctx.set_tuning("spmm_slab", int(slab)); ctx.set_tuning("spmm_kernel", "rows")
            else: ctx.set_tuning("spmm_kernel", "auto")
        for _ in range(3): D.spmm(ctx, a, h, bias, out, act="relu")
        e0 = ctx.event().record()
        for _ in range(args.iters): D.spmm(ctx, a, h, bias, out, act="relu")
        e1 = ctx.event().record()
        ms = e1.elapsed_ms_since(e0) / args.iters
        chk = float(np.abs(out.numpy()[::1009]).sum())
        print(f"round {rnd} slab {slab:>8}: {ms*1e3:9.1f} us  {alg/ms/1e6:8.1f} GB/s  frac {alg/ms/1e6/8000:.3f}  chk {chk:.6g}", flush=True)
if args.pads:
    hh = h.numpy(); del h, out
    for pad in args.pads.split(","):
        keep = [ctx.empty((int(float(pad) * 262144),)) ] if float(pad) > 0 else []
        h2 = ctx.to_device(hh); keep2 = [ctx.empty((int(float(pad) * 262144),))] if float(pad) > 0 else []; o2 = ctx.empty((hb.n, f))
        ctx.set_tuning("spmm_kernel", "auto"); ctx.set_tuning("spmm_slab", 0)
        for _ in range(3): D.spmm(ctx, a, h2, bias, o2, act="relu")
        e0 = ctx.event().record()
        for _ in range(args.iters): D.spmm(ctx, a, h2, bias, o2, act="relu")
        ms = ctx.event().record().elapsed_ms_since(e0) / args.iters
        print(f"pad {pad:>8} MiB  h @{h2.ptr:#x} out @{o2.ptr:#x} diff {(o2.ptr - h2.ptr) / 2**20:10.3f} MiB: {ms*1e3:9.1f} us  frac {alg/ms/1e6/8000:.3f}", flush=True)
        del h2, o2, keep, keep2
ctx.close()
