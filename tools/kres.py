#!/usr/bin/env python
"""Kernel resource usage (VGPRs / spills / occupancy) of one translation unit, per kernel:
    python tools/kres.py magphase_hip.hip [filter] [-D...]"""
import re
import subprocess
import sys

args = sys.argv[1:]
src = args[0]
flt = [a for a in args[1:] if not a.startswith("-")]
defs = [a for a in args[1:] if a.startswith("-")]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-inline-asm", "-c",
       "magphase_amd/csrc/" + src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + defs
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    if flt and not any(f in name for f in flt):
        continue
    print("%-60s VGPR %4s  SGPR %4s  spill %s/%s  scratch %s  occ %s" % (
        name[-60:], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
        r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
