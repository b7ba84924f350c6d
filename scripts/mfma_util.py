#!/usr/bin/env python3
"""MFMA utilisation of the GEMM kernels from the rocprofv3 passes of profile_round.sh.

    python scripts/mfma_util.py gpurun_out/rNN

Per kernel (mean per dispatch): kernel time from the --kernel-trace --stats pass, SQ_VALU_MFMA_BUSY_CYCLES (cycles the
matrix pipes of all SIMDs were busy: 16 per v_mfma_f32_16x16x32_bf16, 64 per v_mfma_f32_32x32x2_f32), GRBM_GUI_ACTIVE
(GPU-busy cycles summed over the 8 XCDs).  Derived:
  clock      = GRBM_GUI_ACTIVE / 8 / kernel time          (what the chip held during the kernel, MI355X_MICROARCH.md DVFS)
  mfma_util  = MFMA_BUSY / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)   (fraction of the kernel's cycles the matrix pipes worked)
  TFLOP/s    = algorithmic flops (2 N Fi Fo, x3 MFMA passes for bf16x3 not counted) / kernel time
  of_peak    = TFLOP/s / dense peak of the instruction (bf16 MFMA 2500, f32 MFMA 157.3; MI355X_MICROARCH.md)
"""
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
N, FI, FO = 1_000_000, 256, 256
flops = 2.0 * N * FI * FO


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


for prec in ("f32", "bf16x3", "bf16"):
    ctr = defaultdict(lambda: defaultdict(list))
    for sub in ("a", "b"):
        for path in glob.glob(os.path.join(root, f"pmc_gemm_{sub}_{prec}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path, newline="")):
                ctr[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = {}
    for path in glob.glob(os.path.join(root, f"trace_gemm_{prec}", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            dur[short(row["Name"])] = float(row["AverageNs"]) * 1e-9
    print(f"== GEMM kernels at N = 1,000,000, 256 x 256, prec {prec}  (algorithmic 131.07 GFLOP per product)")
    for k in sorted(ctr):
        if "gemm" not in k or k not in dur:
            continue
        c = {n: sum(v) / len(v) for n, v in ctr[k].items()}
        t = dur[k]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        clock = gui / 8 / t if t else 0.0
        util = busy / (1024 * gui / 8) if gui else 0.0
        peak = 157.3 if prec == "f32" else 2500.0
        tf = flops / t / 1e12
        waves = c.get("SQ_WAVE_CYCLES", 0.0)
        print(f"  {k:42s} {t * 1e6:8.1f} us  clock {clock / 1e9:4.2f} GHz  MFMA busy {busy:.3e} cyc  mfma_util {util:5.1%}  "
              f"{tf:6.1f} TFLOP/s = {tf / peak:5.1%} of the {peak:g} TF peak  "
              f"wave time: wait(mem/barrier) {c.get('SQ_WAIT_ANY', 0) / max(waves, 1):4.0%} issue-stall {c.get('SQ_WAIT_INST_ANY', 0) / max(waves, 1):4.0%} "
              f"LDS conflicts/active {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):4.0%}")
