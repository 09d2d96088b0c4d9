#!/bin/bash
# visit: random against minimizer-bucketed table placement on E''s real k-mer stream (tools/microbench/placement_probe.hip)
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/microbench/placement_probe.hip -o /tmp/placement_probe 2> gpurun_out/r11k_build.err || { tail -3 gpurun_out/r11k_build.err; exit 1; }
python - <<'PY'
import sys
import numpy as np
sys.path.insert(0, ".")
from autocycler_amd import synth
for name in ("configEprime_k51", "configB_k51"):
    asm = synth.WORKLOADS[name][2]()
    parts = []
    for contigs in asm:
        for _, s in contigs:
            parts.append(np.asarray(s, dtype=np.uint8)); parts.append(np.frombuffer(b"$", dtype=np.uint8))
    np.concatenate(parts).tofile(f"/tmp/{name}.bin")
    print(name, sum(len(p) for p in parts))
PY
for WL in configEprime_k51 configB_k51; do
  timeout 300 /tmp/placement_probe /tmp/$WL.bin 51 31 21 > gpurun_out/r11k_placement_probe_$WL.jsonl 2> gpurun_out/r11k.err; echo "$WL exit $?"; cat gpurun_out/r11k_placement_probe_$WL.jsonl
done
