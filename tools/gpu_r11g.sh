#!/bin/bash
# visit: hybrid upload experiment — every Nth chunk through the pinned ring and a copy, the others stored directly
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "| upload", round(j.get("upload_device_ms", 0) or 0, 3), "insert_k", round(j.get("insert_kernel_ms", 0), 3), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
V="base;AC_UPLOAD_HYBRID=4;AC_UPLOAD_HYBRID=3;AC_UPLOAD_HYBRID=2;base;AC_UPLOAD_HYBRID=4;AC_UPLOAD_HYBRID=8;base"
timeout 600 python tools/ab_knobs.py --steps 12 --host-entry --variants "$V" > gpurun_out/r11g_ab_upload_hybrid_host_entry_configC.jsonl 2> gpurun_out/r11g.err; echo "C exit $?"; show gpurun_out/r11g_ab_upload_hybrid_host_entry_configC.jsonl
uptime; tail -3 gpurun_out/r11g.err
