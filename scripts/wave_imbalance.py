#!/usr/bin/env python3
"""Load imbalance of the aggregation's row-gather kernel: max / mean wave lifetime.
Needs a TUNING build of the library (per-wave s_memrealtime stamps exist only there):
    make -C gcn-string_amd/csrc TUNING=1 OUT=...   or   scripts/build_tuning.sh   ->  scripts/variants/libgcnx_tuning.so
    GCNX_LIB=scripts/variants/libgcnx_tuning.so python scripts/wave_imbalance.py [--workload powerlaw|block1m|ecoli]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gcn-string_amd"))
import numpy as np
import gcnx
from gcnx import device as D, synth
from gcnx.device import DeviceCSR

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="powerlaw")
args = ap.parse_args()
if args.workload == "powerlaw":
    hb = synth.power_law_batch(122, 8192, 256, seed=3, with_x=False, first_graph=0); f = 256
elif args.workload == "block1m":
    sizes, pairs = synth.block_diag_plan()
    hb = synth.block_diag_shard(0, len(sizes), sizes, pairs, 256, seed=2, with_x=False); f = 256
else:
    hb = synth.ecoli_shard(0, 32, 128, seed=1); f = 128
ctx = gcnx.Context(0)
fn = getattr(ctx.lib, "gcnx_tuning_wave_stamps", None)
if fn is None:
    sys.exit("this library has no wave stamps: build with TUNING=1 and point GCNX_LIB at it")
fn.argtypes = [C.c_void_p, C.c_void_p]
vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)
a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, vals, hb.graph_ptr)
h = ctx.to_device(np.random.default_rng(0).standard_normal((hb.n, f), dtype=np.float32))
out = ctx.empty((hb.n, f)); bias = ctx.zeros(f)
ctx.set_tuning("spmm_kernel", "rows")                      # the row gather over every row (what config 5 runs anyway)
for _ in range(3): D.spmm(ctx, a, h, bias, out, act="relu")
nblk = (hb.n + 7) // 8 + 8                                 # at most one workgroup per 8 rows
stamps = ctx.zeros(nblk * 4 * 2, np.uint32)                # uint64 as pairs
ctx._ck(fn(ctx.h, stamps.ptr))
e0 = ctx.event().record()
D.spmm(ctx, a, h, bias, out, act="relu")
ms = ctx.event().record().elapsed_ms_since(e0)
ctx._ck(fn(ctx.h, None))
t = stamps.numpy().view(np.uint64)
t = t[t > 0].astype(np.float64) * 10.0                     # 100 MHz ticks -> ns
deg = np.diff(hb.rowptr)
print(f"{args.workload}: N={hb.n} nnz={hb.nnz} F={f}  max degree {deg.max()}  launch {ms*1e3:.1f} us")
print(f"waves {t.size}: lifetime mean {t.mean()/1e3:.2f} us  median {np.median(t)/1e3:.2f} us  p99 {np.percentile(t, 99)/1e3:.2f} us  "
      f"max {t.max()/1e3:.2f} us   max / mean = {t.max()/t.mean():.2f}   max / launch = {t.max()/(ms*1e6):.3f}")
ctx.close()
