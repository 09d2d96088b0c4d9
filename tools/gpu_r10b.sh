#!/bin/bash
# Round-4 visit b: the alignment cache of the insert (AC_INSERT_ALIGN) — parity digests, then A/B on the diverse and the redundant workloads.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "not config_e_full and not config_d_full" > gpurun_out/r10b_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r10b_pytest.log
export AC_NO_TORCH=1
V="AC_INSERT_ALIGN=0;base;AC_INSERT_WAVES_DIVERSE=16384;AC_INSERT_WAVES_DIVERSE=8192;AC_INSERT_WAVES_DIVERSE=4096;AC_INSERT_ALIGN=0;base"
for W in configEprime_k51 configEmini_k51; do
  timeout 400 python tools/ab_knobs.py --workload $W --steps 6 --variants "$V" > gpurun_out/r10b_ab_$W.jsonl 2> gpurun_out/r10b_ab_$W.err; echo "$W exit $?"
  python - <<PY
import json
for l in open("gpurun_out/r10b_ab_$W.jsonl"):
    j = json.loads(l)
    if "variant" in j: print(j["variant"], round(j.get("ms_median", 0), 3), "insert", round(j.get("insert_kernel_ms", 0), 3), j.get("gfa_md5"), j.get("error", ""))
PY
done
V2="AC_INSERT_ALIGN=0;base;AC_INSERT_ALIGN=0;base"
for W in configB_k51 configDprime_k101; do
  timeout 300 python tools/ab_knobs.py --workload $W --steps 6 --variants "$V2" > gpurun_out/r10b_ab_$W.jsonl 2> gpurun_out/r10b_ab_$W.err; echo "$W exit $?"
  python - <<PY
import json
for l in open("gpurun_out/r10b_ab_$W.jsonl"):
    j = json.loads(l)
    if "variant" in j: print(j["variant"], round(j.get("ms_median", 0), 3), "insert", round(j.get("insert_kernel_ms", 0), 3), j.get("gfa_md5"), j.get("error", ""))
PY
done
V3="base;AC_TABLE_SHIFT=0;AC_PATH_COPY=0,AC_INSERT_ALIGN=0;AC_PATH_COPY=0;base;AC_TABLE_SHIFT=0"
timeout 300 python tools/ab_knobs.py --steps 8 --variants "$V3" > gpurun_out/r10b_ab_configC_k51.jsonl 2> gpurun_out/r10b_ab_configC.err; echo "C exit $?"
python - <<PY
import json
for l in open("gpurun_out/r10b_ab_configC_k51.jsonl"):
    j = json.loads(l)
    if "variant" in j: print(j["variant"], round(j.get("ms_median", 0), 3), "insert", round(j.get("insert_kernel_ms", 0), 3), j.get("gfa_md5"), j.get("error", ""))
PY
