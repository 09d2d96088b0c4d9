#!/bin/bash
# Round-4 visit g: the host side of T_hot — how many packing threads the upload wants on this box.
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread|Core" | head -8
V="base;AC_UPLOAD_THREADS=16;AC_UPLOAD_THREADS=48;AC_UPLOAD_THREADS=64;AC_UPLOAD_THREADS=96;AC_UPLOAD_THREADS=128;base"
timeout 400 python tools/ab_knobs.py --steps 10 --host-entry --variants "$V" > gpurun_out/r10g_ab_upload_threads_host_entry_configC.jsonl 2> gpurun_out/r10g.err; echo "exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/r10g_ab_upload_threads_host_entry_configC.jsonl"):
    j = json.loads(l)
    if "variant" in j: print(j["variant"], "| ms", round(j["ms_median"], 3), "min", round(j["ms_min"], 3), "| upload_device_ms", round(j.get("upload_device_ms", 0), 3), "build", round(j.get("build_ms", 0), 3))
PY
