#!/bin/bash
# visit: the packed upload written straight into device memory by the packing threads (AC_UPLOAD_DIRECT) — parity and A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "| upload", round(j.get("upload_device_ms", 0) or 0, 3), "insert_k", round(j.get("insert_kernel_ms", 0), 3), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -x -q > gpurun_out/r11b_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r11b_pytest.log
export AC_NO_TORCH=1
V="base;AC_UPLOAD_DIRECT=0;base;AC_UPLOAD_DIRECT=0;AC_UPLOAD_THREADS=16;AC_UPLOAD_THREADS=24;AC_UPLOAD_THREADS=48;AC_UPLOAD_THREADS=64;base;AC_UPLOAD_DIRECT=0"
timeout 600 python tools/ab_knobs.py --steps 12 --host-entry --variants "$V" > gpurun_out/r11b_ab_upload_direct_host_entry_configC.jsonl 2> gpurun_out/r11b.err; echo "C exit $?"; show gpurun_out/r11b_ab_upload_direct_host_entry_configC.jsonl
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --host-entry --steps 6 --variants "base;AC_UPLOAD_DIRECT=0;base;AC_UPLOAD_DIRECT=0" > gpurun_out/r11b_ab_upload_direct_host_entry_configEprime.jsonl 2>> gpurun_out/r11b.err; echo "E' exit $?"; show gpurun_out/r11b_ab_upload_direct_host_entry_configEprime.jsonl
tail -3 gpurun_out/r11b.err
