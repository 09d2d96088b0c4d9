#!/bin/bash
# visit: packing threads for the direct upload
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
V="base;AC_UPLOAD_THREADS=48;AC_UPLOAD_THREADS=64;AC_UPLOAD_THREADS=80;AC_UPLOAD_THREADS=96;AC_UPLOAD_THREADS=128;AC_UPLOAD_THREADS=64;AC_UPLOAD_THREADS=96;base;AC_UPLOAD_THREADS=128;AC_UPLOAD_DIRECT=0;AC_UPLOAD_DIRECT=0,AC_UPLOAD_THREADS=64"
timeout 600 python tools/ab_knobs.py --steps 12 --host-entry --variants "$V" > gpurun_out/r11c_ab_upload_direct_threads_host_entry_configC.jsonl 2> gpurun_out/r11c.err; echo "C exit $?"; show gpurun_out/r11c_ab_upload_direct_threads_host_entry_configC.jsonl
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --host-entry --steps 6 --variants "base;AC_UPLOAD_THREADS=64;AC_UPLOAD_THREADS=96;base" > gpurun_out/r11c_ab_upload_direct_threads_host_entry_configEprime.jsonl 2>> gpurun_out/r11c.err; echo "E' exit $?"; show gpurun_out/r11c_ab_upload_direct_threads_host_entry_configEprime.jsonl
nproc; uptime
tail -3 gpurun_out/r11c.err
