#!/bin/bash
# visit p: path entries renumbered on the host (sent in seed numbers right after the walk) — parity and A/B
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
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "HOST_REMAP or REMAP_BLOCK or golden" > gpurun_out/r10p_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r10p_pytest.log
V="base;AC_HOST_REMAP=0;base;AC_HOST_REMAP=0;AC_UPLOAD_THREADS=64;AC_UPLOAD_THREADS=16"
timeout 300 python tools/ab_knobs.py --steps 10 --variants "$V" > gpurun_out/r10p_ab_host_remap_configC_k51.jsonl 2> gpurun_out/r10p.err; echo "C exit $?"; show gpurun_out/r10p_ab_host_remap_configC_k51.jsonl
timeout 300 python tools/ab_knobs.py --steps 10 --host-entry --variants "base;AC_HOST_REMAP=0;base;AC_HOST_REMAP=0" > gpurun_out/r10p_ab_host_remap_host_entry_configC.jsonl 2>> gpurun_out/r10p.err; echo "C host exit $?"; show gpurun_out/r10p_ab_host_remap_host_entry_configC.jsonl
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --steps 6 --variants "base;AC_HOST_REMAP=0;base;AC_HOST_REMAP=0;AC_UPLOAD_THREADS=64" > gpurun_out/r10p_ab_host_remap_configEprime_k51.jsonl 2>> gpurun_out/r10p.err; echo "E' exit $?"; show gpurun_out/r10p_ab_host_remap_configEprime_k51.jsonl
timeout 400 python tools/ab_knobs.py --workload configEmini_k51 --steps 4 --variants "base;AC_HOST_REMAP=1;base;AC_HOST_REMAP=1,AC_UPLOAD_THREADS=64" > gpurun_out/r10p_ab_host_remap_configEmini_k51.jsonl 2>> gpurun_out/r10p.err; echo "mini-E exit $?"; show gpurun_out/r10p_ab_host_remap_configEmini_k51.jsonl
tail -3 gpurun_out/r10p.err
