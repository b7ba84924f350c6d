#!/usr/bin/env python3
"""bench.py -- graphs/sec (fwd+bwd+SGD) of the GCN hot path on N MI355X GPUs, one JSON line.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one train_step (src/scripts/gcn.py:330-340: forward, CCE loss, every gradient, SGD
apply) of the BN-free 2-layer GCNConv model of BASELINE.md section 3 over one synthetic
DisjointLoader batch that is already resident in HBM.  torch is never imported: the launcher
only provides RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*; the data path is libgcnx + RCCL.

Workloads (BASELINE.json configs):
  ecoli    config 2: B=32 E. coli-shaped graphs per GPU, F=128, fp32 (the metric's own config)
  block1m  config 3/4: 1M nodes / 10M entries / F=256 disjoint batch (strong scaling: sharded)
  powerlaw config 5: power-law degrees, max degree 4096
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gcn-string_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)


def make_global_batch(workload, world, scaling):
    from gcnx import synth
    if workload == "ecoli":
        b = 32 * (world if scaling == "weak" else 1)
        hb = synth.ecoli_batch(b, 128, seed=1)
        hidden = 128
    elif workload == "block1m":
        mult = world if scaling == "weak" else 1
        hb = synth.block_diag_batch(1_000_000 * mult, 10_000_000 * mult, 256, seed=2)
        hidden = 256
    elif workload == "powerlaw":
        hb = synth.power_law_batch(122 * (world if scaling == "weak" else 1), 8192, 256, seed=3)
        hidden = 256
    else:
        raise SystemExit(f"unknown workload {workload}")
    hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)      # GCNConv.preprocess (weighted SpMM)
    return hb, hidden


def pmc_traffic(workload):
    """HBM bytes per SpMM call from the committed rocprofv3 PMC passes (profiles/rNN/spmm_pmc.json, collected in
    separate --pmc runs as MI355X_MICROARCH.md prescribes): 2 x FETCH_SIZE (gfx950 counts a wide coalesced read
    at half its bytes) + WRITE_SIZE, KiB -> bytes, summed over the kernels of one gcnx_spmm_csr call."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "spmm_pmc.json")))
    if not files:
        return None
    try:
        tab = json.load(open(files[-1])).get(workload)
        if not tab:
            return None
        tot = 0.0
        for counters in tab.values():
            tot += 2.0 * counters.get("FETCH_SIZE", {}).get("mean_per_dispatch", 0.0)
            tot += counters.get("WRITE_SIZE", {}).get("mean_per_dispatch", 0.0)
        return tot * 1024.0
    except Exception:
        return None


def cpu_baseline(hb, hidden, params_flat, budget_s):
    """Rank 0, N=1 only: the fp32 C restatement (oracle/gcn_oracle.c, OpenMP) of the same step on
    the same batch, on this box's host cores, for about budget_s seconds."""
    from oracle import c_oracle
    threads = min(c_oracle.threads(), int(os.environ.get("GCNX_CPU_THREADS", "16")))   # the box's CPU share for one GPU
    c_oracle.load().orc_set_threads(threads)
    cpu = c_oracle.Gcn2Cpu(hb, hidden, 2, params_flat)
    cpu.step(lr=0.02)                                         # warm-up (page faults, thread pool)
    steps, t0 = 0, time.perf_counter()
    while True:
        cpu.step(lr=0.02)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 2000:
            break
    return {"value": hb.n_graphs * steps / el, "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": f"{steps} train steps of the same batch ({hb.n_graphs} graphs, N={hb.n}, nnz={hb.nnz}, "
                      f"F={hb.f}) in {el:.1f} s; C/OpenMP fp32 restatement, not Spektral/TF (absent)"}


def bench_generalgnn(ctx, args):
    """Secondary line: the reference's live model (GeneralGNN defaults, NetSurfP-width inputs F_in = 16,
    gcn_utills.py:293-300) on the E. coli-shaped batch; forward + CCE + all gradients + SGD, eager launches."""
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GeneralGNN
    assert args.gpus == 1, "this secondary line is single-GPU; the sync-BN multi-GPU step is exercised by tests/ (world_size 2)"
    hb = synth.ecoli_batch(32, 16, seed=1) if args.workload == "ecoli" else synth.block_diag_batch(1_000_000, 10_000_000, 16, seed=2)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    model = GeneralGNN(ctx, 2, activation="softmax", prec=args.prec)
    for _ in range(max(args.warmup, 2)):
        model.train_step(batch, None, lr=0.0002, fetch=False)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.train_step(batch, None, lr=0.0002, fetch=False)
    ctx.sync()
    el = time.perf_counter() - t0
    print(json.dumps({"metric": "graphs/sec (fwd+bwd) GeneralGNN (gcn.py:320 defaults)", "value": hb.n_graphs * args.steps / el,
                      "unit": "graphs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": args.prec, "data": "synthetic",
                      "config": {"workload": f"GeneralGNN(hidden=256, 4 x GeneralConv, BN, PReLU, cat) on {args.workload}: "
                                             f"B={hb.n_graphs}, N={hb.n}, nnz={hb.nnz}, F_in=16", "params": model.n_params}}), flush=True)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ecoli", choices=["ecoli", "block1m", "powerlaw"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--prec", default="f32", choices=["f32", "bf16", "bf16x3"])
    ap.add_argument("--model", default="gcn2", choices=["gcn2", "generalgnn"],
                    help="gcn2 = the BN-free 2-layer GCNConv model of BASELINE.md (default, the metric's model); "
                         "generalgnn = the reference's live model gcn.py:320 (F_in=16, hidden=256; single GPU, eager)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-config3", action="store_true", help="skip the extra config-3 SpMM roofline reading")
    ap.add_argument("--spmm-iters", type=int, default=0, help="SpMM-only launches for the roofline (default 4*steps)")
    args = ap.parse_args()

    import gcnx
    from gcnx import comm as gcomm, device as D, shard, synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2

    rank, local_rank, world = gcomm.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    scaling = args.scaling or ("weak" if args.workload == "ecoli" else "strong")

    ctx = gcnx.Context(local_rank)
    if args.model == "generalgnn":
        return bench_generalgnn(ctx, args)
    comm = gcomm.Communicator(ctx, rank, world)
    hb_global, hidden = make_global_batch(args.workload, world, scaling)
    hb, global_graphs = shard.shard_batch(hb_global, rank, world) if world > 1 else (hb_global, hb_global.n_graphs)

    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    model = GCN2(ctx, 2, hidden=hidden, prec=args.prec, seed=0, comm=comm, use_graph=not args.no_graph)
    model.build(hb.f)
    params0 = np.concatenate([w.ravel() for w in model.get_weights()])
    lr = 0.0002   # the reference's steady-state rate (gcn.py:323: values[-1]); constant so one graph serves

    for _ in range(max(args.warmup, 3)):        # >= 3: eager run, capture, first replay
        model.train_step(batch, None, lr=lr, global_batch=global_graphs, fetch=False)
    comm.barrier()
    ctx.sync()
    ev0 = ctx.event().record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.train_step(batch, None, lr=lr, global_batch=global_graphs, fetch=False)
    ev1 = ctx.event().record()
    ctx.sync()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev1.elapsed_ms_since(ev0)
    elapsed = float(comm.allreduce_host([elapsed], "max")[0])
    loss, acc = model.fetch_metrics(global_graphs)

    # ---- roofline of the dominant hot-path kernel: the GCNConv SpMM (forward shape of layer 1/2)
    iters = args.spmm_iters or 4 * args.steps
    h = ctx.to_device(np.random.default_rng(1).standard_normal((hb.n, hidden), dtype=np.float32))
    out = ctx.empty((hb.n, hidden))
    for _ in range(5):
        D.spmm(ctx, a, h, model.p["b1"], out, act="relu")
    ctx.sync()
    e0 = ctx.event().record()
    for _ in range(iters):
        D.spmm(ctx, a, h, model.p["b1"], out, act="relu")
    e1 = ctx.event().record()
    spmm_ms = e1.elapsed_ms_since(e0) / iters
    alg = synth.spmm_algorithmic_bytes(hb.n, hb.nnz, hidden, weighted=True)
    achieved = alg / (spmm_ms * 1e-3) / 1e9

    # ---- the same kernel at BASELINE config 3 size (1M nodes / 10M entries / F=256), where the launch is long
    # enough for a bandwidth reading; reported next to the primary roofline, never as `value`
    big = None
    if world == 1 and args.workload == "ecoli" and not args.no_config3:
        hb3 = synth.block_diag_batch(with_x=False)
        v3 = synth.gcn_norm_host(hb3.rowptr, hb3.colidx)
        a3 = DeviceCSR.from_host_csr(ctx, hb3.rowptr, hb3.colidx, v3, hb3.graph_ptr)
        h3 = ctx.to_device(np.random.default_rng(2).standard_normal((hb3.n, 256), dtype=np.float32))
        o3 = ctx.empty((hb3.n, 256)); b3 = ctx.zeros(256)
        for _ in range(3):
            D.spmm(ctx, a3, h3, b3, o3, act="relu")
        ctx.sync()
        e0 = ctx.event().record()
        for _ in range(20):
            D.spmm(ctx, a3, h3, b3, o3, act="relu")
        e1 = ctx.event().record()
        ms3 = e1.elapsed_ms_since(e0) / 20
        alg3 = synth.spmm_algorithmic_bytes(hb3.n, hb3.nnz, 256, weighted=True)
        big = {"workload": "config3: N=1,000,000 nnz=10,000,000 F=256 fp32, weighted, bias+relu", "bound": "hbm",
               "achieved": alg3 / (ms3 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": alg3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("block1m"), "algorithmic_bytes": alg3, "avg_launch_us": 1e3 * ms3,
               "launches": 20, "kernels": "spmm_duo_kernel (512- and 1024-thread shapes) + spmm_rows_kernel: one gcnx_spmm_csr call"}
        for t in (a3, h3, o3):
            pass

    if rank == 0:
        small = hb.n < 128 * 1024
        rec = {
            "metric": "graphs/sec (fwd+bwd) on E.coli-sized batches; GCNConv SpMM achieved HBM GB/s",
            "value": global_graphs * args.steps / elapsed, "unit": "graphs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32" if args.prec == "f32" else args.prec,
            "data": "synthetic",
            "config": {"workload": {"ecoli": "config2: E. coli-shaped DisjointLoader batch, B=32 graphs per GPU, F=128, "
                                             "2-layer GCNConv(relu)+GlobalSumPool+Dense softmax, CCE, SGD",
                                    "block1m": "config3/4: 1M-node/10M-entry disjoint batch, F=256, same model",
                                    "powerlaw": "config5: power-law degrees (max 4096), 8192-node graphs, F=256"}[args.workload],
                       "global_graphs": global_graphs, "nodes_per_gpu": hb.n, "nnz_per_gpu": hb.nnz, "features": hb.f,
                       "hidden": hidden, "parallelism": f"dp{world} (graphs sharded, RCCL all-reduce of {model.n_params + 2} fp32)",
                       "hip_graph": not args.no_graph, "gemm_precision": args.prec},
            "device_ms_per_step": dev_ms / args.steps, "final_loss": loss, "final_acc": acc,
            "roofline": {"kernel": ("spmm_rows_kernel" if small else "spmm_duo_kernel(+rows)") +
                                   " (GCNConv aggregation, weighted, bias+relu fused)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.workload), "algorithmic_bytes": alg, "avg_launch_us": 1e3 * spmm_ms, "launches": iters},
        }
        if big is not None:
            rec["roofline_config3"] = big
        if world == 1 and args.cpu_seconds > 0:
            rec["cpu_baseline"] = cpu_baseline(hb, hidden, params0, args.cpu_seconds)
        print(json.dumps(rec), flush=True)
    comm.barrier()
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
