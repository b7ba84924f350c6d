#!/usr/bin/env python3
"""Single-GPU proxy for the N-GPU strong-scaling curve of config 4: runs `bench.py --emulate-rank r --of N`
for every rank r of N in (2, 4, 8) plus the 1-GPU step, and reports T(1) / max_r T_N(r) -- the speed-up an N-GPU run would
reach if the collective cost nothing -- and the same with a stated all-reduce allowance.  Writes one JSON document.
    python scripts/scaling_proxy.py [--workload block1m] [--steps 20] [--out gpurun_out/scaling_proxy.json]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="block1m")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--worlds", default="2,4,8")
ap.add_argument("--allreduce-us", type=float, default=60.0,
                help="allowance for the EXPOSED part of the gradient all-reduce per step (514 KiB fp32 over xGMI: latency-bound; "
                     "SURVEY 8(e) 30-60 us; half of it runs beside layer 1's backward since r3)")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scaling_proxy.json"))
args = ap.parse_args()

def run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--steps", str(args.steps), "--warmup", "3",
           "--cpu-seconds", "0", "--no-config3", "--spmm-iters", "2"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True)
    if out.returncode != 0:
        sys.exit(f"{' '.join(cmd)} failed:\n{out.stderr[-2000:]}")
    return json.loads(out.stdout.strip().splitlines()[-1])

base = run([])
doc = {"workload": base["config"]["workload"], "t1_ms": base["m1_median"]["ms_per_step"], "t1_wall_ms": base["ms_per_step"],
       "allreduce_allowance_us": args.allreduce_us, "worlds": {}}
print(f"1 GPU: {doc['t1_ms']:.3f} ms/step", flush=True)
for w in [int(v) for v in args.worlds.split(",")]:
    ranks = []
    for r in range(w):
        rec = run(["--emulate-rank", str(r), "--of", str(w)])
        ranks.append({"rank": r, "ms": rec["m1_median"]["ms_per_step"], "graphs": rec["emulated"]["shard_graphs"],
                      "cost": rec["emulated"]["shard_cost_nnz_plus_n"]})
    worst = max(x["ms"] for x in ranks)
    mean = sum(x["ms"] for x in ranks) / w
    doc["worlds"][str(w)] = {"ranks": ranks, "max_ms": worst, "mean_ms": mean,
                             "predicted_speedup_no_comm": doc["t1_ms"] / worst,
                             "predicted_speedup_with_allowance": doc["t1_ms"] / (worst + args.allreduce_us * 1e-3),
                             "perfect_split_ms": doc["t1_ms"] / w}
    print(f"{w} ranks: max {worst:.3f} ms (mean {mean:.3f}, T1/{w} = {doc['t1_ms']/w:.3f}) -> predicted speed-up "
          f"{doc['t1_ms']/worst:.2f}x without comm, {doc['t1_ms']/(worst + args.allreduce_us*1e-3):.2f}x with {args.allreduce_us:.0f} us of exposed all-reduce", flush=True)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(doc, open(args.out, "w"), indent=1)
