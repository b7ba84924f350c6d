#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03n; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/a -- python3 scripts/gemm_panel_bench.py --only fwd --ks 1024 --iters 10 > $O/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/b -- python3 scripts/gemm_panel_bench.py --only fwd --ks 1024 --iters 10 > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a", "b"):
    f = sorted(glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_panel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:34s} mean per dispatch {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
