#!/usr/bin/env python3
"""Resource usage of every kernel in a .hip file (VGPRs, spills, scratch, LDS, occupancy): scripts/kres.py csrc/fused.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + root + "/include",
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:],
                     capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: \s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|TotalSGPRs): (\S+)", line)
    if not m:
        if "error" in line: print(line)
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
    cur[k] = v
    if k.startswith("LDS") and flt in cur["name"]:
        name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
        print(f"{name:70s} v={cur.get('VGPRs')} sg={cur.get('TotalSGPRs')} scr={cur.get('ScratchSize [bytes/lane]')} occ={cur.get('Occupancy [waves/SIMD]')} vspill={cur.get('VGPRs Spill')} lds={v}")
