#!/bin/bash
# visit: the other BASELINE workloads on the final code (device entry; median and minimum of the builds after the ramp)
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
: > gpurun_out/r11j_ab_workloads.jsonl
for WL in configB_k51 configDprime_k101 configEprime_k51 configE2_k51 configEmini_k51 configD_k101; do
  timeout 500 python tools/ab_knobs.py --workload $WL --steps 8 --variants "base;base" >> gpurun_out/r11j_ab_workloads.jsonl 2>> gpurun_out/r11j.err; echo "$WL exit $?"
done
python - <<'PY'
import json
for l in open("gpurun_out/r11j_ab_workloads.jsonl"):
    j = json.loads(l)
    if "variant" in j:
        print(j.get("workload"), "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), j.get("gfa_md5", "")[:8], j.get("error", ""), {k: v for k, v in (j.get("stages_ms") or {}).items() if v and v > 0.3 and k != "total_device"})
PY
tail -3 gpurun_out/r11j.err
