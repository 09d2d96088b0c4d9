#!/usr/bin/env python3
"""Extract the reference's hand-written GFA test fixtures into tests/golden/.

Source: /root/reference/src/test_gfa.rs (get_test_gfa_1 .. get_test_gfa_16) — the fixtures the
reference's own unit tests load.  Run in the build container only (the
GPU box has no /root/reference); the extracted .gfa files are committed.
"""
import re
import sys
from pathlib import Path

src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/test_gfa.rs").read_text()
out_dir = Path(__file__).parent
for m in re.finditer(r"pub fn get_test_gfa_(\d+)\(\) -> Vec<String> \{(.*?)\n\}", src, re.S):
    num, body = m.group(1), m.group(2)
    lines = re.findall(r'"((?:[^"\\]|\\.)*)"', body)
    text = "\n".join(l.replace("\\t", "\t") for l in lines) + "\n"
    (out_dir / f"test_gfa_{num}.gfa").write_text(text)
    print(f"test_gfa_{num}.gfa: {len(lines)} lines")
