#!/usr/bin/env python3
"""Print a table of per-kernel register / LDS / scratch usage from hipcc's -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys, os
here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, "..", "rust-ida_amd", "csrc")
out = subprocess.run(["make", "-C", csrc, "resources"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("idahip::", "")}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
print("%-48s %6s %6s %8s %6s %8s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for r in rows:
    print("%-48s %6s %6s %8s %6s %8s" % (r["name"][:48], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
                                      r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
