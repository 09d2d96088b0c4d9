#!/bin/bash
# visit u: the links built on the host (lists sent right after K13) — parity and A/B
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        st = j.get("stages_ms") or {}
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "| upload", round(j.get("upload_device_ms", 0) or 0, 3), "| fin", st.get("finalize"), "d2h", st.get("d2h"), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -x -q > gpurun_out/r10u_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r10u_pytest.log
V="base;AC_HOST_REMAP=0;base;AC_HOST_REMAP=0;AC_UPLOAD_THREADS=64;AC_UPLOAD_THREADS=16"
timeout 300 python tools/ab_knobs.py --steps 10 --variants "$V" > gpurun_out/r10u_ab_host_links_configC_k51.jsonl 2> gpurun_out/r10u.err; echo "C exit $?"; show gpurun_out/r10u_ab_host_links_configC_k51.jsonl
timeout 300 python tools/ab_knobs.py --steps 10 --host-entry --variants "base;AC_HOST_REMAP=0;base;AC_HOST_REMAP=0" > gpurun_out/r10u_ab_host_links_host_entry_configC.jsonl 2>> gpurun_out/r10u.err; echo "C host exit $?"; show gpurun_out/r10u_ab_host_links_host_entry_configC.jsonl
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --steps 6 --variants "base;AC_HOST_REMAP=0;base;AC_HOST_REMAP=0;AC_UPLOAD_THREADS=64" > gpurun_out/r10u_ab_host_links_configEprime_k51.jsonl 2>> gpurun_out/r10u.err; echo "E' exit $?"; show gpurun_out/r10u_ab_host_links_configEprime_k51.jsonl
timeout 400 python tools/ab_knobs.py --workload configEmini_k51 --steps 4 --variants "base;AC_HOST_REMAP=0;base;AC_UPLOAD_THREADS=64" > gpurun_out/r10u_ab_host_links_configEmini_k51.jsonl 2>> gpurun_out/r10u.err; echo "mini-E exit $?"; show gpurun_out/r10u_ab_host_links_configEmini_k51.jsonl
tail -3 gpurun_out/r10u.err
