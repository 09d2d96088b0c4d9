#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of every kernel of graph_build.hip for the W=2 (k=51) instantiations:
    hipcc ... -Rpass-analysis=kernel-resource-usage -c graph_build.hip 2> res.txt ; python tools/kernel_resources.py res.txt"""
import re
import subprocess
import sys

cur = None
rows = {}
for l in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, l)
        if m and cur:
            rows[cur][key] = int(m.group(1))
for name, r in rows.items():
    if "rocprim" in name or re.search(r"Li[134]E", name):
        continue
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(.*", "", d).replace("ac::", "").replace("void ", "")
    print(f"{d[:64]:64s} vgpr={r.get('vgpr')} scratch={r.get('scratch')} occ={r.get('occ')}")
