#!/usr/bin/env python3
"""Static instruction counts of kernels in a hipcc -S --cuda-device-only listing: tools/isa_count.py file.s pattern [pattern...]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
for pat in sys.argv[2:]:
    for n, l in enumerate(lines):
        if l.endswith(":") or ": ;" in l:
            name = l.split(":")[0]
            if name.startswith("_Z") and pat in name:
                ins = []
                for m in lines[n + 1:]:
                    t = m.strip()
                    if t.startswith("s_endpgm"): break
                    if m.startswith("\t") and t and not t.startswith((".", ";")): ins.append(t.split()[0])
                c = Counter(("v_" if i.startswith("v_") else "s_" if i.startswith("s_") else i.split("_")[0] + "_") for i in ins)
                mem = Counter(i for i in ins if i.startswith(("global_", "flat_", "ds_", "buffer_", "scratch_")))
                print(name[:90], len(ins), dict(c), dict(mem.most_common(6)))
