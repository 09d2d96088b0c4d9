#!/bin/bash
# visit s: the path check kernel (RemapFunctor, store = false) is the last kernel of a build now: wavefront block size; t_e2e with / without the host renumbering
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
V="base;AC_REMAP_BLOCK=2048;AC_REMAP_BLOCK=1024;AC_REMAP_BLOCK=512;AC_REMAP_BLOCK=256;base;AC_REMAP_BLOCK=1024;AC_REMAP_BLOCK=512"
timeout 300 python tools/ab_knobs.py --steps 12 --variants "$V" > gpurun_out/r10s_ab_remap_block_configC_k51.jsonl 2> gpurun_out/r10s.err; echo "C exit $?"; show gpurun_out/r10s_ab_remap_block_configC_k51.jsonl
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --steps 6 --variants "base;AC_REMAP_BLOCK=1024;AC_REMAP_BLOCK=512;base" > gpurun_out/r10s_ab_remap_block_configEprime_k51.jsonl 2>> gpurun_out/r10s.err; echo "E' exit $?"; show gpurun_out/r10s_ab_remap_block_configEprime_k51.jsonl
for HR in 1 0 1 0; do
AC_HOST_REMAP=$HR python - <<'PY'
import os, sys, time, json
sys.path.insert(0, ".")
import bench
PY
done 2>/dev/null
for HR in 1 0 1 0; do
  AC_HOST_REMAP=$HR timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --pmc off 2>/dev/null | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); e = j['t_e2e']
print('host_remap $HR t_e2e', e['wall_s'], e['runs_wall_s'], 'write', e['write_s'], 'graph', e['graph_s'], 'cli', (e.get('cli_fresh_process') or {}).get('wall_s'), 't_hot', j['ms_per_step'])"
done
