#!/bin/bash
# Round-4 visit d: the one-launch tail of expand_repeats (AC_EXPAND_TAIL) — parity, then A/B on every workload.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "not config_e_full and not config_d_full" > gpurun_out/r10d_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r10d_pytest.log
export AC_NO_TORCH=1
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        st = j.get("stages_ms", {})
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "| insert_k", round(j.get("insert_kernel_ms", 0), 3),
              "| expand", st.get("expand"), "passes", j.get("simplify_passes"), "tail", j.get("expand_tail_passes"), "levels", j.get("n_levels"), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
V="AC_EXPAND_TAIL=0;base;AC_EXPAND_TAIL=0;base"
for W in configEprime_k51 configEmini_k51 configDprime_k101 configB_k51 configE2_k51; do
  timeout 400 python tools/ab_knobs.py --workload $W --steps 6 --variants "$V" > gpurun_out/r10d_ab_$W.jsonl 2> gpurun_out/r10d_ab_$W.err; echo "$W exit $?"
  show gpurun_out/r10d_ab_$W.jsonl
done
timeout 300 python tools/ab_knobs.py --steps 8 --variants "$V" > gpurun_out/r10d_ab_configC_k51.jsonl 2> gpurun_out/r10d_ab_configC.err; echo "C exit $?"
show gpurun_out/r10d_ab_configC_k51.jsonl
timeout 300 python tools/ab_knobs.py --steps 8 --host-entry --variants "$V" > gpurun_out/r10d_ab_host_entry_configC_k51.jsonl 2> gpurun_out/r10d_ab_host_configC.err; echo "C host exit $?"
show gpurun_out/r10d_ab_host_entry_configC_k51.jsonl
timeout 250 tools/gpu_timeline.sh r10d_configC
grep -c expand_wave gpurun_out/r10d_configC_timeline.txt; grep -n "expand_tail\|DirtyList" gpurun_out/r10d_configC_timeline.txt | head
