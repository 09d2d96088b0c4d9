#!/bin/bash
# visit w: the path-length check on its own queue beside the second renumbering — parity and A/B
export TMPDIR=/tmp
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
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -x -q > gpurun_out/r10w_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r10w_pytest.log
export AC_NO_TORCH=1
V="base;AC_HOST_REMAP=0;base;AC_HOST_REMAP=0;base"
timeout 300 python tools/ab_knobs.py --steps 12 --variants "$V" > gpurun_out/r10w_ab_path_check_aux_stream_configC_k51.jsonl 2> gpurun_out/r10w.err; echo "C exit $?"; show gpurun_out/r10w_ab_path_check_aux_stream_configC_k51.jsonl
timeout 300 python tools/ab_knobs.py --steps 10 --host-entry --variants "base;AC_HOST_REMAP=0;base" > gpurun_out/r10w_ab_path_check_aux_stream_host_entry_configC.jsonl 2>> gpurun_out/r10w.err; echo "C host exit $?"; show gpurun_out/r10w_ab_path_check_aux_stream_host_entry_configC.jsonl
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --steps 6 --variants "base;AC_HOST_REMAP=0;base" > gpurun_out/r10w_ab_path_check_aux_stream_configEprime_k51.jsonl 2>> gpurun_out/r10w.err; echo "E' exit $?"; show gpurun_out/r10w_ab_path_check_aux_stream_configEprime_k51.jsonl
tail -3 gpurun_out/r10w.err
