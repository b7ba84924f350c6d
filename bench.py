#!/usr/bin/env python3
"""bench.py -- graphs/sec (fwd+bwd+SGD) of the GCN hot path on N MI355X GPUs, one JSON line.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts one rank process per GPU itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one train_step (src/scripts/gcn.py:330-340: forward, CCE loss, every gradient, SGD
apply) of the BN-free 2-layer GCNConv model of BASELINE.md section 3 over one synthetic
DisjointLoader batch that is already resident in HBM.  torch is never imported: either launcher
only provides RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*; the data path is libgcnx + RCCL.

Workloads (BASELINE.json configs):
  ecoli    config 2: B=32 E. coli-shaped graphs per GPU, F=128, fp32 (the metric's own config)
  block1m  config 3/4: 1M nodes / 10M entries / F=256 disjoint batch (strong scaling: sharded),
           bf16 MFMA weight GEMMs (plain bf16 operands by default; --prec bf16x3: split-bf16, 2^-18 per operand)
  powerlaw config 5: power-law degrees, max degree 4096
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gcn-string_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)
# config 2 is quoted in fp32; config 3 / 4 state "bf16 MFMA weight GEMM": plain bf16 operands, fp32 accumulate and fp32
# activations (--prec bf16x3 = split-bf16, hi + lo = 16 significand bits per operand at three MFMA passes: every gradient of the
# config-3 step within 3e-6 of an fp64 reference on the same side of the ReLU kinks; --prec f32 = the fp32 MFMA path)
CONFIG_PREC = {"ecoli": "f32", "block1m": "bf16", "powerlaw": "bf16x3"}


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` started plainly spawns the N rank processes itself
# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    """Starts one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / GCNX_RUN_ID in the environment, like
    torch.distributed.run would, plus a fresh run id for the RCCL rendezvous file) BEFORE this process has touched the
    GPU -- it never does.  Rank 0's stdout (the JSON line) is passed through; exits non-zero if any rank does."""
    import socket
    import uuid
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    run_id = uuid.uuid4().hex
    import tempfile
    procs = []
    # rank 0's stdout goes to a temporary FILE, read back after it exits: a pipe that is only read at the end would block
    # the rank (and with it every other rank, until the time limit) once it has written more than the pipe buffer holds
    out0_file = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GCNX_RUN_ID=run_id, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0_file if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + float(os.environ.get("GCNX_BENCH_TIMEOUT", "3000"))
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            p = procs[r]
            code = p.poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        if time.time() > deadline and pending:
            print("bench.py: ranks still running at the launcher's time limit; stopping them", file=sys.stderr)
            for q in pending:
                procs[q].kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    out0_file.seek(0)
    out0 = out0_file.read()
    out0_file.close()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return rc


def launcher_selftest(rank, world):
    """CPU-only plumbing check of the launcher + rendezvous (tests/test_host.py): the ranks exchange a fake 128-byte
    id through the rendezvous file the real Communicator uses; rank 0 prints one JSON line."""
    from gcnx import comm as gcomm
    raw = gcomm.exchange_unique_id(rank, world, timeout_s=60.0, make_id=lambda: bytes(range(128)))
    ok = raw == bytes(range(128))
    # every rank reports through a file so that rank 0 can confirm all of them arrived
    base = gcomm._rendezvous_path() + ".selftest"
    with open(f"{base}.{rank}", "w") as fh:
        fh.write("ok" if ok else "bad")
    if rank == 0:
        t0 = time.time()
        seen = 0
        while time.time() - t0 < 60 and seen < world:
            seen = sum(os.path.exists(f"{base}.{r}") for r in range(world))
            time.sleep(0.02)
        good = all(open(f"{base}.{r}").read() == "ok" for r in range(world)) if seen == world else False
        for r in range(world):
            try:
                os.remove(f"{base}.{r}")
            except OSError:
                pass
        try:
            os.remove(gcomm._rendezvous_path())
        except OSError:
            pass
        print(json.dumps({"selftest": "launcher", "world": world, "ranks_seen": seen, "ok": bool(good),
                          "run_id": os.environ.get("GCNX_RUN_ID")}), flush=True)
        return 0 if good else 1
    return 0 if ok else 1


# ---------------------------------------------------------------------------------------------------------------
# workloads: every rank builds only its own shard (graph g comes from its own random stream)
# ---------------------------------------------------------------------------------------------------------------
def make_shard(workload, rank, world, scaling):
    """This rank's graphs of the global batch + the global graph count.  Sizes of ALL graphs are generated everywhere
    (cheap; the cost-balanced partition needs them), edges / features only for the rank's own range."""
    from gcnx import shard, synth
    if workload == "ecoli":
        b = 32 * (world if scaling == "weak" else 1)
        sizes = synth.ecoli_sizes(b, seed=1)
        # cost model of shard.partition_graphs with the entry count estimated from the size (mean degree ~ 15)
        bounds = shard.partition_by_cost(sizes * 16.0, world)
        hb = synth.ecoli_shard(int(bounds[rank]), int(bounds[rank + 1]), 128, seed=1)
        hidden = 128
    elif workload == "block1m":
        mult = world if scaling == "weak" else 1
        sizes, pairs = synth.block_diag_plan(1_000_000 * mult, 10_000_000 * mult, seed=2)
        bounds = shard.partition_by_cost(sizes + 2 * pairs + sizes, world)       # nnz_g + n_g
        hb = synth.block_diag_shard(int(bounds[rank]), int(bounds[rank + 1]), sizes, pairs, 256, seed=2)
        b, hidden = len(sizes), 256
    elif workload == "powerlaw":
        b = 122 * (world if scaling == "weak" else 1)
        bounds = shard.partition_by_cost(np.ones(b), world)                      # same-size graphs
        hb = synth.power_law_batch(int(bounds[rank + 1] - bounds[rank]), 8192, 256, seed=3, first_graph=int(bounds[rank]))
        hidden = 256
    else:
        raise SystemExit(f"unknown workload {workload}")
    hb.vals = synth.gcn_norm_host(hb.rowptr, hb.colidx)      # GCNConv.preprocess (weighted SpMM)
    return hb, hidden, b


def pmc_traffic(workload):
    """HBM bytes per SpMM call from the committed rocprofv3 PMC passes (profiles/rNN/spmm_pmc.json, collected in
    separate --pmc runs as MI355X_MICROARCH.md prescribes): 2 x FETCH_SIZE (gfx950 counts a wide coalesced read
    at half its bytes) + WRITE_SIZE, KiB -> bytes, summed over the kernels of one gcnx_spmm_csr call.  A committed
    figure of an earlier profiling run of the same kernels, NOT measured in this run: see `traffic_source`."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "spmm_pmc.json")))
    if not files:
        return None, None
    try:
        tab = json.load(open(files[-1])).get(workload)
        if not tab:
            return None, None
        tot = 0.0
        for counters in tab.values():
            tot += 2.0 * counters.get("FETCH_SIZE", {}).get("mean_per_dispatch", 0.0)
            tot += counters.get("WRITE_SIZE", {}).get("mean_per_dispatch", 0.0)
        return tot * 1024.0, os.path.relpath(files[-1], ROOT) + " (static: committed rocprofv3 --pmc passes, not this run)"
    except Exception:
        return None, None


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


def cpu_baseline_scipy(hb, hidden, params_flat, budget_s):
    """the NumPy/SciPy restatement of the same step (scipy.sparse CSR @ dense for the
    aggregation, NumPy/BLAS GEMMs), fp32, the threads NumPy's BLAS takes by itself.  A second CPU reference next to
    the C/OpenMP port; never the optimisation target."""
    import scipy.sparse as sp
    f, h, c = hb.f, hidden, 2
    a = sp.csr_matrix((hb.vals.astype(np.float32), hb.colidx, hb.rowptr), shape=(hb.n, hb.n))
    off = 0
    p = {}
    for k, shp in (("w1", (f, h)), ("b1", (h,)), ("w2", (h, h)), ("b2", (h,)), ("w3", (h, c)), ("b3", (c,))):
        n = int(np.prod(shp)); p[k] = params_flat[off:off + n].reshape(shp).astype(np.float32).copy(); off += n
    seg = np.repeat(np.arange(hb.n_graphs), np.diff(hb.graph_ptr))
    pool = sp.csr_matrix((np.ones(hb.n, np.float32), (seg, np.arange(hb.n))), shape=(hb.n_graphs, hb.n))
    x, y = hb.x, hb.y

    def step(lr):
        y1 = np.maximum(a @ (x @ p["w1"]) + p["b1"], 0)
        y2 = np.maximum(a @ (y1 @ p["w2"]) + p["b2"], 0)
        pooled = pool @ y2
        z = pooled @ p["w3"] + p["b3"]
        z = z - z.max(1, keepdims=True)
        pr = np.exp(z); pr /= pr.sum(1, keepdims=True)
        dl = (pr - y) / np.float32(hb.n_graphs)
        g = {"w3": pooled.T @ dl, "b3": dl.sum(0)}
        dz = (pool.T @ (dl @ p["w3"].T)) * (y2 > 0)
        g["b2"] = dz.sum(0); dh = a.T @ dz
        g["w2"] = y1.T @ dh
        dz = (dh @ p["w2"].T) * (y1 > 0)
        g["b1"] = dz.sum(0); dh = a.T @ dz
        g["w1"] = x.T @ dh
        for k in p:
            p[k] -= np.float32(lr) * g[k].astype(np.float32)
    step(0.0002)
    steps, t0 = 0, time.perf_counter()
    while True:
        step(0.0002); steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 500:
            break
    return {"value": hb.n_graphs * steps / el, "unit": "graphs/s", "kind": "port (NumPy/SciPy, BASELINE.md C2)",
            "sample": f"{steps} train steps in {el:.1f} s; scipy.sparse csr @ dense + NumPy BLAS, fp32"}


# Under rocprofv3 (LD_PRELOADed tool library, ROCPROF_* variables) a deep queue of hipGraphLaunch calls crashes the profiled
# process: SIGSEGV inside hipGraphLaunch, in the tool's interception of the queue writes.  Cause established in round 4 with a
# product-free reproducer (scripts/micro/graph_trace_repro.hip, results in profiles/r04/graph_trace_repro_summary.txt): a
# plain HIP program that captures 150 small kernel nodes and launches the graph 400 times without waiting dies under
# `rocprofv3 --kernel-trace` with the same frames (same page offsets) as the product's crash of round 3, runs clean without
# the tracer, clean with a synchronisation every 8 launches, and clean with a 5-node graph launched 8000 times -- a defect of
# the tool's dispatch tracking for large graphs enqueued deeply, not a lifetime problem of the library's captured graphs.  The
# timed loops below never wait for the GPU, so a profiled run drains the queue (every 8 steps; every step for the 150-node
# GeneralGNN graph); the JSON line says "profiled": true; kernel durations, which is what such a run is for, are unaffected.
PROFILED = "rocprofiler-sdk" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)


def drain(ctx, k, every=8):
    if PROFILED and k % every == every - 1:
        ctx.sync()


def env_knobs():
    """GCNX_* tuning knobs present in the environment (diagnostics: some change kernel selection)."""
    allowed = {"GCNX_RUN_ID", "GCNX_CPU_THREADS", "GCNX_BENCH_TIMEOUT", "GCNX_LIB", "GCNX_ROCTX"}   # (ROCTX: trace markers only)
    return {k: v for k, v in os.environ.items() if k.startswith("GCNX_") and k not in allowed}


def bench_generalgnn(ctx, args):
    """Secondary line: the reference's live model (GeneralGNN defaults, NetSurfP-width inputs F_in = 16,
    gcn_utills.py:293-300) on the E. coli-shaped batch; forward + CCE + all gradients + SGD."""
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GeneralGNN
    assert args.gpus == 1, "this secondary line is single-GPU; the sync-BN multi-GPU step is exercised by tests/ (world_size 2)"
    hb = synth.ecoli_shard(0, 32, 16, seed=1) if args.workload == "ecoli" else synth.block_diag_batch(1_000_000, 10_000_000, 16, seed=2)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    model = GeneralGNN(ctx, 2, activation="softmax", prec=args.prec)
    for _ in range(max(args.warmup, 3)):
        model.train_step(batch, None, lr=0.0002, fetch=False)
    ctx.sync()
    burn_steps, t_burn = 0, time.perf_counter()           # untimed burn-in, as in main()
    while (time.perf_counter() - t_burn) * 1e3 < args.burn_in_ms:
        model.train_step(batch, None, lr=0.0002, fetch=False)
        ctx.sync()
        burn_steps += 1
    t0 = time.perf_counter()
    for k in range(args.steps):
        model.train_step(batch, None, lr=0.0002, fetch=False)
        drain(ctx, k, 1)
    ctx.sync()
    el = time.perf_counter() - t0
    print(json.dumps({"metric": "graphs/sec (fwd+bwd) GeneralGNN (gcn.py:320 defaults)", "value": hb.n_graphs * args.steps / el,
                      "unit": "graphs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "burn_in": {"steps": burn_steps, "ms": args.burn_in_ms},
                      "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": args.prec, "data": "synthetic",
                      "config": {"workload": f"GeneralGNN(hidden=256, 4 x GeneralConv, BN, PReLU, cat) on {args.workload}: "
                                             f"B={hb.n_graphs}, N={hb.n}, nnz={hb.nnz}, F_in=16", "params": model.n_params}}), flush=True)
    ctx.close()


def generalgnn_bytes(layers, n, b, nnz, hidden, mp):
    """Per-op compulsory HBM bytes of one GeneralGNN training step: every launchable operation of the step reads
    each of its operands once and writes its result once, fp32, perfect reuse INSIDE an operation and none across -- the
    figure `generalgnn.*.hbm_frac` is taken on.  Dense: x, W in, z out; BatchNorm + activation: z in, y out; aggregation:
    CSR + h in, rows out; pool; backward: the transposed aggregation, BatchNorm backward (dy, z in, dz out -- one read of
    each although the batch statistics of the gradient need two), dW (x, dz in), dX (dz, W in, dx accumulated: read +
    write, first contribution write only)."""
    tot = 0
    wcat = hidden * (mp + 1)
    for i, L in enumerate(layers):
        r = b if L["group"] == "post" else n
        fi, fo = L["fi"], L["fo"]
        tot += 4 * (r * fi + fi * fo + r * fo)            # forward product
        tot += 8 * r * fo                                 # BatchNorm + activation
        tot += 12 * r * fo                                # ... backward
        tot += 4 * (r * fi + r * fo + fi * fo)            # dW
        if i > 0:
            tot += 4 * (r * fo + fi * fo) + (8 if L["group"] == "gnn" else 4) * r * fi     # dX (accumulated into the skip slices)
        if L["group"] == "gnn":
            tot += 2 * (4 * (n + 1) + 4 * nnz + 8 * n * hidden)                              # aggregation, forward and transposed
    tot += 2 * 4 * (n * wcat + b * wcat)                  # pool forward / backward
    return tot


def generalgnn_cpu_baseline(hb, budget_s):
    """The fp32 NumPy restatement (oracle/gcn_oracle.py: general_gnn_loss_and_grads -- BLAS products, np.add.at aggregation)
    of the same GeneralGNN step on the same batch, on this box's host cores: a reported baseline, "port", bounded sample."""
    from oracle import gcn_oracle as O
    rng = np.random.default_rng(0)
    layers = O.general_gnn_init(rng, hb.f, 2, dtype=np.float32)
    x, y = hb.x.astype(np.float32), hb.y.astype(np.float32)
    csr = (hb.rowptr.astype(np.int64), hb.colidx.astype(np.int64), None)
    steps, t0 = 0, time.perf_counter()
    while True:
        O.general_gnn_loss_and_grads(layers, x, csr, hb.graph_ptr, y)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 50:
            break
    return {"value": hb.n_graphs * steps / el, "unit": "graphs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{steps} forward + loss + gradient evaluations (no update) of the same batch in {el:.1f} s; NumPy fp32 restatement "
                      f"(BLAS threads as NumPy finds them), not Spektral/TF (absent)"}


def generalgnn_extra(ctx, steps=60, cpu_seconds=6.0):
    """Extra key of the default line: the reference's LIVE model -- GeneralGNN(2, activation="softmax"),
    gcn.py:320, NetSurfP-width inputs (F_in = 16, gcn_utills.py:293-300) -- on the same E. coli-shaped batch: forward + CCE +
    every gradient + SGD from one captured HIP graph, in exact fp32 and with the Dense products on the split-bf16 panel
    kernels (bf16x3: 2^-18 per operand).  flops = the step's dense products (forward, dX, dW); the aggregation adds none."""
    from gcnx import synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GeneralGNN
    hb = synth.ecoli_shard(0, 32, 16, seed=1)
    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, None, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    out = {"workload": f"GeneralGNN(hidden=256, 4 x GeneralConv, BN, PReLU, cat) on the config-2 batch with F_in=16: B={hb.n_graphs}, "
                       f"N={hb.n}, nnz={hb.nnz}"}
    keep = []
    for prec in ("f32", "bf16x3"):
        # (captured step; under a profiler this EXTRA runs eagerly: as the third and fourth model of the process its graph replays
        # still die inside the tracer -- profiles/r04/graph_trace_repro_summary.txt, LOG "Round 4" -- although every step is
        # synchronised; a stand-alone traced `--model generalgnn` run captures and replays without trouble)
        model = GeneralGNN(ctx, 2, activation="softmax", prec=prec, use_graph=not PROFILED)
        for _ in range(5):
            model.train_step(batch, None, lr=0.0002, fetch=False)
        ctx.sync()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) < 0.06:            # the same untimed burn-in as the main line
            model.train_step(batch, None, lr=0.0002, fetch=False)
            ctx.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            model.train_step(batch, None, lr=0.0002, fetch=False)
            drain(ctx, k, 1)      # (profiled runs only: see PROFILED -- a 150-node graph is not enqueued deeply under the tracer)
        ctx.sync()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        flops = 0
        for i, L in enumerate(model.layers):
            rows = hb.n_graphs if L["group"] == "post" else hb.n
            flops += 2 * rows * L["fi"] * L["fo"] * (2 if i == 0 else 3)
        peak = 157.3e12 if prec == "f32" else 2.5e15 / 3.0   # fp32 MFMA; bf16 MFMA at three products per multiply
        nbytes = generalgnn_bytes(model.layers, hb.n, hb.n_graphs, hb.nnz, model.hidden, model.mp)
        out[prec] = {"ms_per_step": ms, "graphs_per_s": hb.n_graphs / (ms * 1e-3), "flops": flops,
                     "frac_of_mfma_peak": flops / (ms * 1e-3) / peak, "params": model.n_params,
                     "compulsory_bytes": nbytes, "hbm_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "bound": "hbm (per-op compulsory bytes, generalgnn_bytes(); the products are far below the MFMA peak)"}
        keep.append(model)          # (captured graphs are destroyed with the process)
    if cpu_seconds > 0:
        out["cpu_baseline"] = generalgnn_cpu_baseline(hb, cpu_seconds)
    return out


CONFIG3_STEP_1GPU_MS = {"value": 2.84, "source": "BENCH_r03.json config3_step (round 3, one MI355X, bf16 operands)"}


def config4_plan(world):
    """The strong-scaling cut of BASELINE config 4: per-rank (graphs, nodes, entries) of the config-3 batch as
    shard.partition_by_cost deals it to `world` ranks -- from the graphs' sizes alone, no graph is built."""
    from gcnx import shard, synth
    sizes, pairs = synth.block_diag_plan(1_000_000, 10_000_000, seed=2)
    bounds = shard.partition_by_cost(sizes + 2 * pairs + sizes, world)
    return [{"rank": r, "graphs": int(bounds[r + 1] - bounds[r]), "nodes": int(sizes[bounds[r]:bounds[r + 1]].sum()),
             "entries": int((sizes + 2 * pairs)[bounds[r]:bounds[r + 1]].sum())} for r in range(world)], int(len(sizes))


def config4_extra(ctx, comm, rank, world, lr, use_graph, steps=20, burn=30):
    """Extra key of an N > 1 line: BASELINE config 4 -- the config-3 batch (1M nodes / 10M entries /
    F = 256, hidden 256, bf16 weight-GEMM operands) STRONG-scaled over the N ranks: every rank builds and holds its shard
    (contiguous graph range, cost-balanced), the step is forward + CCE + all gradients + the RCCL all-reduce + SGD, loss
    normalised by the global batch.  Timed like the main line (barrier + sync on both sides, max over ranks).  The one-GPU
    reference it is compared with is measured in the same run: rank 0 runs the whole batch's step alone while the other
    ranks wait at a barrier (their GPUs idle); the committed round-3 figure is reported next to it."""
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2
    prec = CONFIG_PREC["block1m"]

    def timed(hb, global_graphs, cm, k_burn, k_steps):
        a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
        batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
        m = GCN2(ctx, 2, hidden=256, prec=prec, seed=0, comm=cm, use_graph=use_graph)
        m.build(hb.f)
        for k in range(3 + k_burn):                      # eager, capture, replay; then the untimed burn-in (a fixed count:
            m.train_step(batch, None, lr=lr, global_batch=global_graphs, fetch=False)   # every step holds a collective)
            drain(ctx, k)
        if cm is not None:
            cm.barrier()
        ctx.sync()
        t0 = time.perf_counter()
        for k in range(k_steps):
            m.train_step(batch, None, lr=lr, global_batch=global_graphs, fetch=False)
            drain(ctx, k)
        ctx.sync()
        if cm is not None:
            cm.barrier()
        el = time.perf_counter() - t0
        loss, acc = m.fetch_metrics(global_graphs)
        return el, loss, m, batch

    hb, _, global_graphs = make_shard("block1m", rank, world, "strong")
    el, loss, m_keep, b_keep = timed(hb, global_graphs, comm, burn, steps)
    el = float(comm.allreduce_host([el], "max")[0])
    ms = 1e3 * el / steps
    nodes = comm.allreduce_host([hb.n, hb.nnz], "max")
    out = {"workload": f"config4: the config-3 batch (1M nodes / 10M entries / F=256, {global_graphs} graphs) sharded over {world} GPUs "
                       f"(contiguous graph ranges balanced on entries + nodes), hidden=256, weight GEMMs {prec}, fwd + CCE + all "
                       f"gradients + RCCL all-reduce + SGD", "scaling": "strong", "n_gpus": world, "steps": steps, "burn_in_steps": burn,
           "ms_per_step": ms, "graphs_per_s": global_graphs / (ms * 1e-3), "nodes_per_s": 1_000_000 / (ms * 1e-3),
           "largest_shard": {"nodes": int(nodes[0]), "entries": int(nodes[1])}, "final_loss": loss,
           "all_reduce": "inside the step graph" if m_keep._comm_in_graph() else "eager between two graphs", "hip_graph": use_graph}
    # the one-GPU step of the same batch, on rank 0, the other GPUs idle
    one_ms = None
    if rank == 0:
        hb1, _, _ = make_shard("block1m", 0, 1, "strong")
        el1, loss1, m1, b1 = timed(hb1, global_graphs, None, burn, steps)
        one_ms = 1e3 * el1 / steps
        out["one_gpu_same_run"] = {"ms_per_step": one_ms, "graphs_per_s": global_graphs / (one_ms * 1e-3), "final_loss": loss1,
                                   "what": "rank 0 alone on the whole batch, the other ranks waiting at a barrier"}
        out["speedup_vs_one_gpu_same_run"] = one_ms / ms
        out["speedup_vs_committed_1gpu"] = CONFIG3_STEP_1GPU_MS["value"] / ms
        out["committed_1gpu"] = CONFIG3_STEP_1GPU_MS
    comm.barrier()
    return out


def time_spmm(ctx, D, a, h, bias, out, iters):
    for _ in range(5):
        D.spmm(ctx, a, h, bias, out, act="relu")
    ctx.sync()
    e0 = ctx.event().record()
    for _ in range(iters):
        D.spmm(ctx, a, h, bias, out, act="relu")
    e1 = ctx.event().record()
    return e1.elapsed_ms_since(e0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ecoli", choices=["ecoli", "block1m", "powerlaw"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--prec", default=None, choices=["f32", "bf16", "bf16x3"],
                    help="GEMM arithmetic; default per workload: ecoli f32 (config 2 is an fp32 config), block1m / powerlaw "
                         "bf16x3 (config 3's bf16 MFMA weight GEMM as split-bf16: three MFMA passes, 2^-18 relative per operand)")
    ap.add_argument("--model", default="gcn2", choices=["gcn2", "generalgnn"],
                    help="gcn2 = the BN-free 2-layer GCNConv model of BASELINE.md (default, the metric's model); "
                         "generalgnn = the reference's live model gcn.py:320 (F_in=16, hidden=256; single GPU)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches everywhere")
    ap.add_argument("--graph", action="store_true", help="one captured HIP graph per step everywhere (default: the model's own choice -- "
                                                          "eager for the five-launch step of config 2 in one process, captured otherwise)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--scipy-seconds", type=float, default=4.0, help="budget of the NumPy/SciPy baseline C2 (0 = skip)")
    ap.add_argument("--no-config3", action="store_true", help="skip the extra config-3 SpMM roofline reading")
    ap.add_argument("--no-generalgnn", action="store_true", help="skip the extra GeneralGNN (gcn.py:320) reading of the default line")
    ap.add_argument("--spmm-iters", type=int, default=0, help="SpMM-only launches for the roofline (default 4*steps)")
    ap.add_argument("--burn-in-ms", type=float, default=60.0,
                    help="untimed repetitions of the step before the timed region until this much wall time has passed "
                         "(0 = none); reported as burn_in in the JSON line")
    ap.add_argument("--allow-knobs", action="store_true", help="run although GCNX_* tuning knobs are set (they are recorded)")
    ap.add_argument("--emulate-rank", type=int, default=None, metavar="R",
                    help="single-GPU proxy of an N-GPU run (VERDICT r2, next 3): build rank R's shard of the workload as --of N "
                         "ranks would cut it (shard.partition_by_cost, loss normalised by the GLOBAL batch) and run its step on "
                         "this one GPU without a communicator; scripts/scaling_proxy.py takes the max over R against the 1-GPU step")
    ap.add_argument("--of", type=int, default=8, metavar="N", help="world size --emulate-rank cuts the workload for")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--selftest-config4", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-config4", action="store_true", help="N > 1: skip the extra config-4 (strong-scaled 1M-node batch) reading")
    args = ap.parse_args()
    if args.prec is None:
        args.prec = CONFIG_PREC[args.workload]

    # ---- N > 1 started plainly: become the launcher (nothing below has touched the GPU yet) ----------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    from gcnx import comm as gcomm
    rank, local_rank, world = gcomm.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"or plainly as `python bench.py --gpus {args.gpus}`")
    if args.selftest_launcher:
        sys.exit(launcher_selftest(rank, world))
    if args.selftest_config4:
        # CPU-only check (tests/test_host.py) that an N > 1 line carries the config4_step key and what it is cut into: the ranks
        # meet through the launcher's rendezvous like the real run, rank 0 prints the line's skeleton with the shard plan
        if rank == 0:
            plan, b = config4_plan(world)
            print(json.dumps({"selftest": "config4", "n_gpus": world, "scaling_of_main_line": args.scaling or "weak",
                              "config4_step": {"scaling": "strong", "n_gpus": world, "global_graphs": b, "shards": plan,
                                               "ms_per_step": None, "graphs_per_s": None}}), flush=True)
        sys.exit(launcher_selftest(rank, world))
    knobs = env_knobs()
    if knobs and not args.allow_knobs:
        raise SystemExit(f"bench.py: tuning knobs set in the environment ({knobs}); numbers taken with them are not the "
                         f"product's.  Unset them or pass --allow-knobs (they are then recorded in the JSON line).")

    import gcnx
    from gcnx import device as D, synth
    from gcnx.device import DeviceCSR, Segments
    from gcnx.models import DeviceBatch, GCN2

    scaling = args.scaling or ("weak" if args.workload == "ecoli" else "strong")
    ctx = gcnx.Context(local_rank)
    if args.model == "generalgnn":
        return bench_generalgnn(ctx, args)
    comm = gcomm.Communicator(ctx, rank, world)
    if args.emulate_rank is not None:
        if world != 1 or not 0 <= args.emulate_rank < args.of:
            raise SystemExit("--emulate-rank R --of N is a single-process mode with 0 <= R < N")
        hb, hidden, global_graphs = make_shard(args.workload, args.emulate_rank, args.of, scaling)
    else:
        hb, hidden, global_graphs = make_shard(args.workload, rank, world, scaling)

    a = DeviceCSR.from_host_csr(ctx, hb.rowptr, hb.colidx, hb.vals, hb.graph_ptr)
    batch = DeviceBatch(ctx, ctx.to_device(hb.x), a, Segments(ctx, hb.graph_ptr), ctx.to_device(hb.y))
    model = GCN2(ctx, 2, hidden=hidden, prec=args.prec, seed=0, comm=comm, use_graph=False if args.no_graph else (True if args.graph else "auto"))
    model.build(hb.f)
    params0 = np.concatenate([w.ravel() for w in model.get_weights()])
    lr = 0.0002   # the reference's steady-state rate (gcn.py:323: values[-1]); constant so one graph serves

    for _ in range(max(args.warmup, 3)):        # >= 3: eager run, capture, first replay
        model.train_step(batch, None, lr=lr, global_batch=global_graphs, fetch=False)
    # Burn-in (untimed, reported in the JSON line): a step here is 0.1 ms, so W warm-up steps are well under a millisecond
    # of GPU work and the chip is still coming out of idle when the timed region starts -- the first few hundred steps
    # then run 5-8 % slower than the rest (20 steps after 5 warm-up steps: 0.110 ms/step; after 50 ms of work: 0.1025;
    # a 2000-step run: 0.1017).  A training run is minutes long: the steady state is the quantity of interest, so the same
    # step is repeated until `--burn-in-ms` of wall time have passed, then the K timed steps follow as the contract says.
    # (The count is agreed between the ranks -- every step holds a collective: 5 probe steps, max over ranks.)
    # The contract-cold figure next to the steady-state one: K steps timed right behind the W warm-up
    # steps, before any burn-in -- what `value` would be with --burn-in-ms 0.
    comm.barrier(); ctx.sync()
    t_cold = time.perf_counter()
    for k in range(args.steps):
        model.train_step(batch, None, lr=lr, global_batch=global_graphs, fetch=False)
        drain(ctx, k)
    ctx.sync(); comm.barrier()
    cold_ms = 1e3 * float(comm.allreduce_host([time.perf_counter() - t_cold], "max")[0]) / args.steps
    burn_steps = 0
    if args.burn_in_ms > 0:
        ctx.sync(); t_burn = time.perf_counter()
        for _ in range(5):
            model.train_step(batch, None, lr=lr, global_batch=global_graphs, fetch=False)
        ctx.sync()
        per = float(comm.allreduce_host([(time.perf_counter() - t_burn) / 5], "max")[0])
        burn_steps = 5 + int(min(max(args.burn_in_ms * 1e-3 / per - 5, 0), 5000))
        for k in range(burn_steps - 5):
            model.train_step(batch, None, lr=lr, global_batch=global_graphs, fetch=False)
            drain(ctx, k)
    comm.barrier()
    ctx.sync()
    evs = [ctx.event() for _ in range(args.steps + 1)]     # created outside the timed region
    evs[0].record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.train_step(batch, None, lr=lr, global_batch=global_graphs, fetch=False)
        evs[i + 1].record()                     # per-step HIP events on the ctx stream (median)
        drain(ctx, i)
    ctx.sync()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    per_step = np.array([evs[i + 1].elapsed_ms_since(evs[i]) for i in range(args.steps)])
    dev_ms = float(per_step.sum())
    med_ms = float(np.median(per_step))
    elapsed = float(comm.allreduce_host([elapsed], "max")[0])
    med_ms = float(comm.allreduce_host([med_ms], "max")[0])
    loss, acc = model.fetch_metrics(global_graphs)

    # ---- roofline of the dominant hot-path kernel: the GCNConv SpMM (forward shape of layer 1/2)
    iters = args.spmm_iters or 4 * args.steps
    h = ctx.to_device(np.random.default_rng(1).standard_normal((hb.n, hidden), dtype=np.float32))
    out = ctx.empty((hb.n, hidden))
    spmm_ms = time_spmm(ctx, D, a, h, model.p["b1"], out, iters)
    alg = synth.spmm_algorithmic_bytes(hb.n, hb.nnz, hidden, weighted=True)
    achieved = alg / (spmm_ms * 1e-3) / 1e9

    # ---- small-feature regime (config 2): the step does not launch gcnx_spmm_csr at all -- every GCNConv is ONE launch,
    # aggregation + dense product (csrc/fused.hip).  Its launch time and bytes, measured the same way, next to the
    # stand-alone aggregation above (which stays the `roofline` entry: it is the kernel the metric names).
    fused = None
    if getattr(model, "_fused", None) and model._fused(batch):
        s_buf = ctx.empty((hb.n, hb.f)); y_buf = ctx.empty((hb.n, hidden))
        for _ in range(5):
            D.gcn_conv_fwd(ctx, a, batch.x, model.p["w1"], model.p["b1"], y_buf, act="relu", s=s_buf, prec=args.prec)
        e0 = ctx.event().record()
        for _ in range(iters):
            D.gcn_conv_fwd(ctx, a, batch.x, model.p["w1"], model.p["b1"], y_buf, act="relu", s=s_buf, prec=args.prec)
        e1 = ctx.event().record()
        f_ms = e1.elapsed_ms_since(e0) / iters
        f_alg = 4 * (hb.n + 1) + 8 * hb.nnz + 4 * hb.n * hb.f * 2 + 4 * hb.n * hidden + 4 * hb.f * hidden
        fused = {"kernel": "gcn_conv_fused_kernel: gather + " + ("fp32 MFMA (16x16x4)" if args.prec == "f32" else "split-bf16 MFMA (16x16x32 x 3)") +
                           " dense + bias/ReLU, S = A X saved; what the "
                           "training step launches per GCNConv at this config instead of gcnx_gemm + gcnx_spmm_csr",
                 "bound": "hbm", "achieved": f_alg / (f_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": f_alg / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": f_alg,
                 "avg_launch_us": 1e3 * f_ms, "launches": iters,
                 "note": "latency regime: the gather moves nnz x 4F bytes through the CUs' L1 paths (64 B/clk each), the dense "
                         "product is fp32-MFMA-bound per CU; neither phase is near HBM's rate at 37 MB per launch"}

    # ---- the same kernel at BASELINE config 3 size (1M nodes / 10M entries / F=256), where the launch is long
    # enough for a bandwidth reading; reported next to the primary roofline, never as `value`
    big = step3 = big5 = None
    if world == 1 and args.workload == "ecoli" and not args.no_config3:
        sizes3, pairs3 = synth.block_diag_plan()
        hb3 = synth.block_diag_shard(0, len(sizes3), sizes3, pairs3, 256, seed=2, with_x=True)
        v3 = synth.gcn_norm_host(hb3.rowptr, hb3.colidx)
        a3 = DeviceCSR.from_host_csr(ctx, hb3.rowptr, hb3.colidx, v3, hb3.graph_ptr)
        # the whole config-3 train step (what `--workload block1m` times), so that the driver's run holds it too (VERDICT r2,
        # weak 12): same model, weight GEMMs in the precision config 3 states, one captured HIP graph per step
        batch3 = DeviceBatch(ctx, ctx.to_device(hb3.x), a3, Segments(ctx, hb3.graph_ptr), ctx.to_device(hb3.y))
        m3 = GCN2(ctx, 2, hidden=256, prec=CONFIG_PREC["block1m"], seed=0, use_graph=not args.no_graph)
        m3.build(hb3.f)
        for _ in range(3):
            m3.train_step(batch3, None, lr=lr, global_batch=hb3.n_graphs, fetch=False)
        ctx.sync()
        t3 = time.perf_counter()
        while time.perf_counter() - t3 < 0.1:                # the same kind of untimed burn-in as the main line
            m3.train_step(batch3, None, lr=lr, global_batch=hb3.n_graphs, fetch=False)
            ctx.sync()
        k3 = 20
        t3 = time.perf_counter()
        for k in range(k3):
            m3.train_step(batch3, None, lr=lr, global_batch=hb3.n_graphs, fetch=False)
            drain(ctx, k)
        ctx.sync()
        ms_step3 = 1e3 * (time.perf_counter() - t3) / k3
        loss3, acc3 = m3.fetch_metrics(hb3.n_graphs)
        step3 = {"workload": f"config3: 1M-node/10M-entry disjoint batch ({hb3.n_graphs} graphs), F=256, hidden=256, the headline's model, "
                             f"weight GEMMs {CONFIG_PREC['block1m']}, fwd + CCE + all gradients + SGD", "ms_per_step": ms_step3,
                 "graphs_per_s": hb3.n_graphs / (ms_step3 * 1e-3), "nodes_per_s": hb3.n / (ms_step3 * 1e-3), "steps": k3,
                 "hip_graph": not args.no_graph, "final_loss": loss3,
                 "what": "the step `python bench.py --workload block1m` times as its value, measured inside the default run"}
        # (m3 and its captured graphs live until the process ends: destroying graphs in the middle of a profiled run has crashed the tracer)
        h3 = ctx.to_device(np.random.default_rng(2).standard_normal((hb3.n, 256), dtype=np.float32))
        o3 = ctx.empty((hb3.n, 256)); b3 = ctx.zeros(256)
        time_spmm(ctx, D, a3, h3, b3, o3, 40)              # untimed: 20 ms of the same launches first
        ms3 = time_spmm(ctx, D, a3, h3, b3, o3, 60)        # (60 launches = 33 ms: the block1m line times 4 x steps of them)
        alg3 = synth.spmm_algorithmic_bytes(hb3.n, hb3.nnz, 256, weighted=True)
        tr3, src3 = pmc_traffic("block1m")
        big = {"workload": "config3: N=1,000,000 nnz=10,000,000 F=256 fp32, weighted, bias+relu", "bound": "hbm",
               "achieved": alg3 / (ms3 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": alg3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tr3, "traffic_source": src3,
               "algorithmic_bytes": alg3, "avg_launch_us": 1e3 * ms3,
               "launches": 60, "kernels": "all kernels of one gcnx_spmm_csr call (tile tiers + row chunks)"}
        # ... and with bf16 features (both dtypes reported): gcnx_spmm_csr_bf16, bf16 in / bf16 out
        hb16 = D.to_bf16(ctx, h3); ob16 = ctx.empty((hb3.n, 256), np.uint16)
        for _ in range(3):
            D.spmm_bf16(ctx, a3, hb16, b3, ob16, act="relu")
        ctx.sync()
        e0 = ctx.event().record()
        for _ in range(20):
            D.spmm_bf16(ctx, a3, hb16, b3, ob16, act="relu")
        ms16 = ctx.event().record().elapsed_ms_since(e0) / 20
        alg16 = 4 * (hb3.n + 1) + 8 * hb3.nnz + 2 * 2 * hb3.n * 256
        big_bf16 = {"workload": "config3 with bf16 features: N=1,000,000 nnz=10,000,000 F=256, weighted, bias+relu, bf16 in / bf16 out, "
                                "fp32 accumulation", "kernel": "spmm_bf16_kernel (row gather, 8 features per lane)", "bound": "hbm",
                    "achieved": alg16 / (ms16 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": alg16 / (ms16 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": alg16,
                    "avg_launch_us": 1e3 * ms16, "launches": 20,
                    "note": "a measured variant: the models keep fp32 activations (LOG.md section 7, item 7)"}
        # ... and the aggregation at BASELINE config 5 (power-law degrees up to 4096, 122 graphs of 8192 nodes), the skewed case
        hb5 = synth.power_law_batch(122, 8192, 256, seed=3, with_x=False, first_graph=0)
        a5 = DeviceCSR.from_host_csr(ctx, hb5.rowptr, hb5.colidx, synth.gcn_norm_host(hb5.rowptr, hb5.colidx), hb5.graph_ptr)
        h5 = ctx.to_device(np.random.default_rng(3).standard_normal((hb5.n, 256), dtype=np.float32))
        o5 = ctx.empty((hb5.n, 256))
        time_spmm(ctx, D, a5, h5, b3, o5, 20)
        ms5 = time_spmm(ctx, D, a5, h5, b3, o5, 40)
        alg5 = synth.spmm_algorithmic_bytes(hb5.n, hb5.nnz, 256, weighted=True)
        tr5, src5 = pmc_traffic("powerlaw")
        big5 = {"workload": f"config5: power-law degrees (max {int(np.diff(hb5.rowptr).max())}), {hb5.n_graphs} graphs of 8192 nodes, "
                            f"N={hb5.n} nnz={hb5.nnz} F=256 fp32, weighted, bias+relu", "bound": "hbm",
                "achieved": alg5 / (ms5 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg5 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": tr5, "traffic_source": src5, "algorithmic_bytes": alg5, "avg_launch_us": 1e3 * ms5, "launches": 40,
                "kernels": "row gather + hub-row segments (rows of more than 256 entries on their own workgroups) of one gcnx_spmm_csr call"}

    gnn_extra = None
    if world == 1 and args.workload == "ecoli" and args.emulate_rank is None and not args.no_generalgnn:
        gnn_extra = generalgnn_extra(ctx, cpu_seconds=min(args.cpu_seconds, 6.0))
    cfg4 = None
    if world > 1 and args.workload == "ecoli" and not args.no_config4 and not args.no_config3:
        cfg4 = config4_extra(ctx, comm, rank, world, lr, not args.no_graph)
    if rank == 0:
        small = hb.n < 128 * 1024
        tr, src = pmc_traffic(args.workload)
        rec = {
            "metric": "graphs/sec (fwd+bwd) on E.coli-sized batches; GCNConv SpMM achieved HBM GB/s",
            "value": global_graphs * args.steps / elapsed, "unit": "graphs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32" if args.prec == "f32" else args.prec,
            "data": "synthetic",
            "config": {"workload": {"ecoli": "config2: E. coli-shaped DisjointLoader batch, B=32 graphs per GPU, F=128, fp32, "
                                             "2-layer GCNConv(relu)+GlobalSumPool+Dense softmax, CCE (from-logits form of tf.function), SGD",
                                    "block1m": f"config3/4: 1M-node/10M-entry disjoint batch, F=256, same model, weight GEMMs {args.prec} "
                                               "(bf16x3 = split-bf16 on the bf16 MFMA, 2^-18 per operand; bf16 = plain bf16 operands)",
                                    "powerlaw": f"config5: power-law degrees (max 4096), 8192-node graphs, F=256, weight GEMMs {args.prec}"}[args.workload],
                       "global_graphs": global_graphs, "nodes_per_gpu": hb.n, "nnz_per_gpu": hb.nnz, "features": hb.f,
                       "hidden": hidden, "parallelism": f"dp{world} (graphs sharded, RCCL all-reduce of {model.n_params + 2} fp32"
                                                        f"{' inside the step graph' if world > 1 and model._comm_in_graph() else ''})",
                       "hip_graph": bool(model.use_graph), "gemm_precision": args.prec, "cce": model.cce_train,
                       "activation_storage": ("bf16 for the tensors only bf16-operand weight GEMMs read (S1, Y1, dH2, dZ1: results "
                                              "bit-identical to fp32 storage), fp32 elsewhere" if (model._bufs or {}).get("act16") else "fp32")},
            "burn_in": {"steps": burn_steps, "ms": args.burn_in_ms,
                        "what": "untimed repetitions of the same step after the W warm-up steps and before the timed region (clock / "
                                "cache steady state; the timed region is exactly `steps` full steps)"},
            "cold": {"ms_per_step": cold_ms, "graphs_per_s": global_graphs / (cold_ms * 1e-3),
                     "what": "the same K steps timed directly behind the W warm-up steps, before the burn-in (the figure --burn-in-ms 0 gives)"},
            "device_ms_per_step": dev_ms / args.steps,
            "m1_median": {"ms_per_step": med_ms, "graphs_per_s": global_graphs / (med_ms * 1e-3),
                          "what": "median of per-step HIP-event times on the ctx stream, max over ranks (SURVEY 8(d) M1); "
                                  "`value` is the wall-clock figure the bench contract defines"},
            "final_loss": loss, "final_acc": acc, "profiled": PROFILED,
            "roofline": {"kernel": ("spmm_rows_kernel" if small else "spmm tile kernel(s) + row chunks") +
                                   " (GCNConv aggregation, weighted, bias+relu fused)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": tr, "traffic_source": src, "algorithmic_bytes": alg, "avg_launch_us": 1e3 * spmm_ms, "launches": iters},
        }
        if args.emulate_rank is not None:
            # `value` stays this process's own rate (its shard's graphs per second); the proxy's reading is ms_per_step
            shard_graphs = hb.n_graphs
            rec["value"] = shard_graphs * args.steps / elapsed
            rec["emulated"] = {"rank": args.emulate_rank, "of": args.of, "scaling": scaling, "shard_graphs": shard_graphs,
                               "global_graphs": global_graphs, "shard_cost_nnz_plus_n": int(hb.nnz + hb.n),
                               "what": "rank R's shard of an N-rank run on ONE GPU, no communicator: the step time a rank would "
                                       "need before its all-reduce; NOT a multi-GPU measurement"}
        if knobs:
            rec["env_knobs"] = knobs
        if fused is not None:
            rec["roofline_step_kernel"] = fused
        if gnn_extra is not None:
            rec["generalgnn"] = gnn_extra
        if cfg4 is not None:
            rec["config4_step"] = cfg4
        if big is not None:
            rec["roofline_config3"] = big
            rec["roofline_config3_bf16"] = big_bf16
            rec["config3_step"] = step3
            rec["roofline_config5"] = big5
        if world == 1 and args.cpu_seconds > 0 and args.emulate_rank is None:
            rec["cpu_baseline"] = cpu_baseline(hb, hidden, params0, args.cpu_seconds)
            if args.scipy_seconds > 0:
                rec["cpu_baseline_scipy"] = cpu_baseline_scipy(hb, hidden, params0, args.scipy_seconds)
        print(json.dumps(rec), flush=True)
    comm.barrier()
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
