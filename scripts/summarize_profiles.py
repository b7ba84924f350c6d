#!/usr/bin/env python3
"""Condenses one profile_round.sh output directory: bench lines, top kernels, per-launch SpMM PMC traffic."""
import collections, csv, glob, json, os, shutil, sys
d = sys.argv[1]
pmc_json = {}
for w in ("ecoli", "block1m", "powerlaw", "generalgnn"):
    p = os.path.join(d, f"bench_{w}.json")
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                r = json.loads(line)
                rf = r.get("roofline")
                if rf is None:
                    print(f"[bench {w}] {r['value']:.0f} graphs/s  {r['ms_per_step']:.3f} ms/step  ({r['config']['workload'][:80]})"); continue
                print(f"[bench {w}] {r['value']:.0f} graphs/s  {r['ms_per_step']:.3f} ms/step  {rf.get('kernel', 'SpMM')[:40]} {rf.get('avg_launch_us', 0):.1f} us "
                      f"{rf['achieved']:.0f} {rf['unit']} frac {rf['frac']:.3f}")
                if "cold" in r: print(f"        cold (right after warm-up) {r['cold']}")
                if "generalgnn" in r: print(f"        generalgnn {r['generalgnn']}")
                if "roofline_config3" in r:
                    b = r["roofline_config3"]; print(f"        config3 SpMM {b['avg_launch_us']:.1f} us {b['achieved']:.0f} GB/s frac {b['frac']:.3f}")
                if "cpu_baseline" in r:
                    c = r["cpu_baseline"]; print(f"        cpu_baseline {c['value']:.1f} graphs/s on {c['cores']} threads ({c['kind']})")
    for f in sorted(glob.glob(os.path.join(d, f"trace_{w}", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]:
        shutil.copy(f, os.path.join(d, f"kernel_stats_{w}.csv"))
        print(f"[kernel stats {w}]")
        for i, row in enumerate(csv.DictReader(open(f))):
            if i < 12:
                name = row["Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
                print(f"   {name:60s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:9.1f} us  {row['Percentage']:>6s}%")
    traffic = {}
    for kind in ("fetch", "write"):
        for f in sorted(glob.glob(os.path.join(d, f"pmc_{kind}_{w}", "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                if "spmm_" in k:
                    short = k.split("spmm_")[1].split("(")[0]
                    agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, v in agg.items():
                for c, x in v.items():
                    traffic.setdefault(k, {})[c] = sum(x) / len(x)
                    pmc_json.setdefault(w, {}).setdefault(k, {})[c] = {"mean_per_dispatch": sum(x) / len(x), "dispatches": len(x)}
    if traffic:
        print(f"[SpMM PMC per launch, {w}]  (FETCH_SIZE/WRITE_SIZE in KiB as reported; gfx950: FETCH_SIZE under-reports wide coalesced reads 2x)")
        tot_f = tot_w = 0.0
        for k, v in traffic.items():
            print("   ", k[:60], {c: f"{x:.4g}" for c, x in v.items()})
            tot_f += v.get("FETCH_SIZE", 0); tot_w += v.get("WRITE_SIZE", 0)
        print(f"    sum over the call's kernels: FETCH {tot_f*1024/1e6:.1f} MB (x2 corrected {2*tot_f*1024/1e6:.1f} MB)  WRITE {tot_w*1024/1e6:.1f} MB")
if pmc_json:
    json.dump(pmc_json, open(os.path.join(d, "spmm_pmc.json"), "w"), indent=1)
