#!/bin/bash
# visit l: streaming stores in the host packers (AC_PACK_NT), the link under host memory load, where the pinned ring lives
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "| upload_device_ms", round(j.get("upload_device_ms", 0) or 0, 3), "insert_k", round(j.get("insert_kernel_ms", 0), 3), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
hipcc -O3 --offload-arch=gfx950 tools/microbench/upload_probe.hip -o /tmp/upload_probe -Lautocycler_amd -lautocycler_hip -Wl,-rpath,$PWD/autocycler_amd -pthread 2> gpurun_out/r10l_build.err || { tail -5 gpurun_out/r10l_build.err; exit 1; }
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo " <- GPU numa nodes"
for NT in 0 1; do
  AC_PACK_NT=$NT timeout 300 /tmp/upload_probe > gpurun_out/r10l_upload_probe_nt$NT.jsonl 2> gpurun_out/r10l_probe.err; echo "probe nt=$NT exit $?"
  cat gpurun_out/r10l_upload_probe_nt$NT.jsonl
done
V="base;AC_PACK_NT=1;base;AC_PACK_NT=1;AC_PACK_NT=1,AC_UPLOAD_THREADS=48;AC_PACK_NT=1,AC_UPLOAD_THREADS=24;AC_PACK_NT=1,AC_UPLOAD_THREADS=16"
timeout 400 python tools/ab_knobs.py --steps 10 --host-entry --variants "$V" > gpurun_out/r10l_ab_pack_nt_host_entry_configC.jsonl 2> gpurun_out/r10l_n.err; echo "ab exit $?"; show gpurun_out/r10l_ab_pack_nt_host_entry_configC.jsonl
