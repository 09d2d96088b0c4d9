#!/bin/bash
# Round-4 visit j: streaming stores for the fills (AC_FILL_NT), packing threads pinned to the text's NUMA node (AC_UPLOAD_NUMA).
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
numactl --hardware 2>/dev/null | head -6
for FN in 0 1; do
  AC_FILL_NT=$FN timeout 200 python tools/ab_knobs.py --steps 10 --variants "base;base" > gpurun_out/r10j_ab_fill_nt${FN}_configC_k51.jsonl 2> gpurun_out/r10j_f.err; echo "fill_nt=$FN exit $?"; show gpurun_out/r10j_ab_fill_nt${FN}_configC_k51.jsonl
done
for FN in 0 1; do
  AC_FILL_NT=$FN timeout 200 python tools/ab_knobs.py --workload configEprime_k51 --steps 6 --variants "base;base" > gpurun_out/r10j_ab_fill_nt${FN}_configEprime_k51.jsonl 2> gpurun_out/r10j_f.err; echo "E' fill_nt=$FN exit $?"; show gpurun_out/r10j_ab_fill_nt${FN}_configEprime_k51.jsonl
done
V="base;AC_UPLOAD_NUMA=1;AC_UPLOAD_NUMA=1,AC_UPLOAD_THREADS=48;AC_UPLOAD_NUMA=1,AC_UPLOAD_THREADS=64;AC_UPLOAD_NUMA=1,AC_UPLOAD_THREADS=24;base;AC_UPLOAD_NUMA=1"
timeout 400 python tools/ab_knobs.py --steps 10 --host-entry --variants "$V" > gpurun_out/r10j_ab_upload_numa_host_entry_configC.jsonl 2> gpurun_out/r10j_n.err; echo "numa exit $?"; show gpurun_out/r10j_ab_upload_numa_host_entry_configC.jsonl
