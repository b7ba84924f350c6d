#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
   usage: hipcc ... -c file.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python scripts/kres.py [filter]"""
import re, subprocess, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur, rows = None, []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            pass
        name = re.sub(r"\(.*", "", name).replace("(anonymous namespace)::", "").replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    if flt in r["name"]:
        print(f'{r["name"]:60s} vgpr={r.get("VGPRs","?"):>4s} agpr={r.get("AGPRs","?"):>3s} scratch={r.get("ScratchSize [bytes/lane]","?"):>4s} '
              f'occ={r.get("Occupancy [waves/SIMD]","?"):>2s} lds={r.get("LDS Size [bytes/block]","?")}')
