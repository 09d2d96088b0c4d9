#!/bin/bash
# Round-5 visit g: hand-written primitives, second version (batched look-back, scan state pool, fewer digits) — tests, A/B timings, timeline.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -k "primitives" > gpurun_out/r12g_pytest_prims.log 2>&1; echo "prims exit $?"; tail -3 gpurun_out/r12g_pytest_prims.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r12g_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r12g_pytest.log
export AC_NO_TORCH=1
timeout 300 python tools/ab_knobs.py --steps 14 --variants "base;base" > gpurun_out/r12g_ab_configC.jsonl 2> gpurun_out/r12g.err; echo "C exit $?"
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --steps 8 --variants "base;base" > gpurun_out/r12g_ab_configEprime.jsonl 2>> gpurun_out/r12g.err; echo "E' exit $?"
python - <<'PY'
import json
for f in ("r12g_ab_configC", "r12g_ab_configEprime"):
    for l in open(f"gpurun_out/{f}.jsonl"):
        j = json.loads(l)
        if "variant" in j:
            print(f, j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "launches", j.get("launches"), "readbacks", j.get("readbacks"), j.get("stages_ms"), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/r12g_prof -o stats -- python $OLDPWD/tools/ab_knobs.py --steps 6 --variants "base" > $OLDPWD/gpurun_out/r12g_prof_ab.json 2> $OLDPWD/gpurun_out/r12g_prof.err; echo "rocprof exit $?"; cd $OLDPWD
DB=$(find gpurun_out/r12g_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r12g_kernel_stats_configC.csv && head -24 gpurun_out/r12g_kernel_stats_configC.csv
[ -n "$DB" ] && python tools/rocpd_launches.py $DB "" 400 0 > gpurun_out/r12g_timeline_all.txt 2>/dev/null
find gpurun_out/r12g_prof -type f -size +8M -delete
tail -n 3 gpurun_out/r12g.err
