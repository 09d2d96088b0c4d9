#!/bin/bash
# Round-4 visit c: occupancy of the latency-bound kernels (AC_INSERT_OCC, AC_OCC_BOOST) x alignment cache, A/B on every workload.
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
V="AC_INSERT_ALIGN=0,AC_OCC_BOOST=0;AC_INSERT_ALIGN=0,AC_INSERT_OCC=8,AC_OCC_BOOST=0;AC_INSERT_ALIGN=1,AC_INSERT_OCC=8,AC_OCC_BOOST=0;AC_INSERT_ALIGN=0,AC_INSERT_OCC=8;AC_INSERT_ALIGN=1,AC_INSERT_OCC=8;AC_INSERT_ALIGN=0,AC_OCC_BOOST=0"
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        st = j.get("stages_ms", {})
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "| insert_k", round(j.get("insert_kernel_ms", 0), 3),
              "| degree", st.get("degree"), "links", st.get("links"), "paths", st.get("paths"), "expand", st.get("expand"), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
for W in configEprime_k51 configEmini_k51 configDprime_k101 configB_k51; do
  timeout 400 python tools/ab_knobs.py --workload $W --steps 6 --variants "$V" > gpurun_out/r10c_ab_$W.jsonl 2> gpurun_out/r10c_ab_$W.err; echo "$W exit $?"
  show gpurun_out/r10c_ab_$W.jsonl
done
VC="AC_INSERT_ALIGN=0,AC_OCC_BOOST=0;AC_INSERT_ALIGN=0,AC_INSERT_OCC=8,AC_OCC_BOOST=0;AC_INSERT_ALIGN=0,AC_INSERT_OCC=8;AC_INSERT_ALIGN=0,AC_OCC_BOOST=0;AC_INSERT_ALIGN=0,AC_INSERT_OCC=8"
timeout 300 python tools/ab_knobs.py --steps 8 --variants "$VC" > gpurun_out/r10c_ab_configC_k51.jsonl 2> gpurun_out/r10c_ab_configC.err; echo "C exit $?"
show gpurun_out/r10c_ab_configC_k51.jsonl
timeout 300 python tools/ab_knobs.py --steps 8 --host-entry --variants "$VC" > gpurun_out/r10c_ab_host_entry_configC_k51.jsonl 2> gpurun_out/r10c_ab_host_configC.err; echo "C host exit $?"
show gpurun_out/r10c_ab_host_entry_configC_k51.jsonl
