#!/bin/bash
# Round-4 visit i: launch geometry of the final renumbering (AC_REMAP_BLOCK), kernel statistics of config C, configs[4] at full size.
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        st = j.get("stages_ms", {})
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), {k: st.get(k) for k in ("finalize", "d2h", "paths", "expand")}, j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
V="base;AC_REMAP_BLOCK=2048;AC_REMAP_BLOCK=1024;AC_REMAP_BLOCK=512;AC_REMAP_BLOCK=256;base"
timeout 300 python tools/ab_knobs.py --steps 8 --variants "$V" > gpurun_out/r10i_ab_remap_block_configC_k51.jsonl 2> gpurun_out/r10i_c.err; echo "C exit $?"; show gpurun_out/r10i_ab_remap_block_configC_k51.jsonl
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --steps 6 --variants "$V" > gpurun_out/r10i_ab_remap_block_configEprime_k51.jsonl 2> gpurun_out/r10i_e.err; echo "E' exit $?"; show gpurun_out/r10i_ab_remap_block_configEprime_k51.jsonl
timeout 250 tools/gpu_timeline.sh r10i_configC
head -45 gpurun_out/r10i_configC_kernel_stats.csv
