#!/bin/bash
# Round-5 visit d: the library without rocPRIM — the hand-written scan / radix sort on the device, parity, the bench lines, kernel statistics.
export TMPDIR=/tmp
mkdir -p gpurun_out
ls -la autocycler_amd/libautocycler_hip.so
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -k "primitives or verify" > gpurun_out/r12d_pytest_prims.log 2>&1; echo "prims exit $?"; tail -3 gpurun_out/r12d_pytest_prims.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q > gpurun_out/r12d_pytest.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r12d_pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "digest_equals" > gpurun_out/r12d_pytest_digests.log 2>&1; echo "digests exit $?"; tail -3 gpurun_out/r12d_pytest_digests.log
export AC_NO_TORCH=1
timeout 300 python tools/ab_knobs.py --steps 14 --variants "base;base" > gpurun_out/r12d_ab_configC.jsonl 2> gpurun_out/r12d.err; echo "C exit $?"
timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --steps 8 --variants "base;base" > gpurun_out/r12d_ab_configEprime.jsonl 2>> gpurun_out/r12d.err; echo "E' exit $?"
python - <<'PY'
import json
for f in ("r12d_ab_configC", "r12d_ab_configEprime"):
    for l in open(f"gpurun_out/{f}.jsonl"):
        j = json.loads(l)
        if "variant" in j:
            print(f, j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), j.get("stages_ms"), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
unset AC_NO_TORCH
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/r12d_prof -o stats -- python $OLDPWD/bench.py --steps 3 --warmup 1 --init-builds 0 --no-cpu-baseline --no-e2e --pmc off > $OLDPWD/gpurun_out/r12d_prof_bench.json 2> $OLDPWD/gpurun_out/r12d_prof.err; echo "rocprof exit $?"; cd $OLDPWD
DB=$(find gpurun_out/r12d_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r12d_kernel_stats_configC.csv && head -30 gpurun_out/r12d_kernel_stats_configC.csv
find gpurun_out/r12d_prof -type f -size +8M -delete
tail -n 3 gpurun_out/r12d.err
