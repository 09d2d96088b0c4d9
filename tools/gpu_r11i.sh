#!/bin/bash
# visit: the host threads of the path renumbering woken while the device sorts (instead of after FinalMetaFunctor) — A/B
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        st = j.get("stages_ms") or {}
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "| fin", st.get("finalize"), "d2h", st.get("d2h"), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
V="base;AC_REMAP_LATE=1;base;AC_REMAP_LATE=1;base;AC_REMAP_LATE=1"
timeout 300 python tools/ab_knobs.py --steps 14 --variants "$V" > gpurun_out/r11i_ab_remap_early_start_configC_k51.jsonl 2> gpurun_out/r11i.err; echo "C exit $?"; show gpurun_out/r11i_ab_remap_early_start_configC_k51.jsonl
timeout 300 python tools/ab_knobs.py --steps 10 --host-entry --variants "base;AC_REMAP_LATE=1;base;AC_REMAP_LATE=1" > gpurun_out/r11i_ab_remap_early_start_host_entry_configC.jsonl 2>> gpurun_out/r11i.err; echo "C host exit $?"; show gpurun_out/r11i_ab_remap_early_start_host_entry_configC.jsonl
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --steps 6 --variants "base;AC_REMAP_LATE=1;base;AC_REMAP_LATE=1" > gpurun_out/r11i_ab_remap_early_start_configEprime_k51.jsonl 2>> gpurun_out/r11i.err; echo "E' exit $?"; show gpurun_out/r11i_ab_remap_early_start_configEprime_k51.jsonl
tail -3 gpurun_out/r11i.err
