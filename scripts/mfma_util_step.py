#!/usr/bin/env python3
"""MFMA utilisation of every GEMM kernel of a whole training step (r4: the r3 panel / bf16-storage kernels).
    python scripts/mfma_util_step.py PMC_DIR_A PMC_DIR_B TRACE_DIR
PMC_DIR_A holds a --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... pass, PMC_DIR_B a --pmc GRBM_GUI_ACTIVE ... pass, TRACE_DIR a
--kernel-trace --stats pass of the SAME bench command.  Per kernel (mean per dispatch):
  clock     = GRBM_GUI_ACTIVE / 8 XCDs / kernel time
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): fraction of the kernel's cycles the matrix pipes worked"""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


ctr = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:3]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            ctr[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur, calls = {}, {}
for path in glob.glob(os.path.join(sys.argv[3], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        dur[short(row["Name"])] = float(row["AverageNs"]) * 1e-9
        calls[short(row["Name"])] = int(row["Calls"])
print(f"{'kernel':58s} {'calls':>6s} {'us':>8s} {'GHz':>5s} {'MFMA busy':>9s} {'wait':>5s} {'LDS confl':>9s}")
for k in sorted(ctr, key=lambda k: -dur.get(k, 0) * calls.get(k, 0)):
    if "gemm" not in k or k not in dur:
        continue
    c = {n: sum(v) / len(v) for n, v in ctr[k].items()}
    t, gui, busy = dur[k], c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    waves = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    print(f"{k[:58]:58s} {calls[k]:6d} {t * 1e6:8.1f} {gui / 8 / t / 1e9 if t else 0:5.2f} {busy / (1024 * gui / 8) if gui else 0:9.1%} "
          f"{c.get('SQ_WAIT_ANY', 0) / waves:5.0%} {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):9.0%}")
