#!/bin/bash
# Round-5 visit j: counters calibrated on random access, stamped PMC traffic for C / E' / mini-E / D, their bench lines, kernel statistics.
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/pmc_calibrate_random.sh r12j > gpurun_out/r12j_cal.log 2>&1; tail -40 gpurun_out/r12j_cal.log
cp gpurun_out/r12j_pmc_calibration_random.json profiles/pmc_calibration_random.json
PMC_TIMEOUT=200 bash tools/pmc_lean.sh r12j base > gpurun_out/r12j_pmc_C.log 2>&1
PMC_TIMEOUT=300 bash tools/pmc_lean.sh r12jE base configEprime_k51 > gpurun_out/r12j_pmc_E.log 2>&1
PMC_TIMEOUT=400 bash tools/pmc_lean.sh r12jM base configEmini_k51 > gpurun_out/r12j_pmc_M.log 2>&1
PMC_TIMEOUT=900 bash tools/pmc_lean.sh r12jD base configD_k101 > gpurun_out/r12j_pmc_D.log 2>&1
for P in ":configC_k51:pmc_traffic.json" "E:configEprime_k51:pmc_traffic_configEprime_k51.json" "M:configEmini_k51:pmc_traffic_configEmini_k51.json" "D:configD_k101:pmc_traffic_configD_k101.json"; do
  S=${P%%:*}; R=${P#*:}; WL=${R%%:*}; OUT=${R#*:}
  NT=$(python tools/workload_n_text.py $WL)
  echo "workload $WL n_text $NT"
  python tools/pmc_traffic.py gpurun_out/r12j${S}_pmc_FETCH_SIZE.csv gpurun_out/r12j${S}_pmc_WRITE_SIZE.csv $NT r12j${S} 7 $WL profiles/pmc_calibration_random.json > gpurun_out/$OUT && cp gpurun_out/$OUT profiles/$OUT
done
python - <<'PY'
import json
for f in ("pmc_traffic.json", "pmc_traffic_configEprime_k51.json", "pmc_traffic_configEmini_k51.json", "pmc_traffic_configD_k101.json"):
    try:
        d = json.load(open("gpurun_out/" + f)); print(f, d["source_hash"], "raw", round(d["traffic_raw"] / 1e9, 3), "x2", round(d["traffic_streaming_x2"] / 1e9, 3), "applied", round(d["traffic_bytes_per_build"] / 1e9, 3), "factor", d["fetch_correction"], d["calibration"]["fetch_raw_over_known"])
    except Exception as e: print(f, "failed", e)
PY
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r12j_bench_configC.json 2> gpurun_out/r12j_bench.err; echo "bench C exit $?"
timeout 600 python bench.py --workload configEprime_k51 --steps 10 --warmup 3 --no-e2e > gpurun_out/r12j_bench_configEprime.json 2>> gpurun_out/r12j_bench.err; echo "bench E' exit $?"
timeout 900 python bench.py --workload configEmini_k51 --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > gpurun_out/r12j_bench_configEmini.json 2>> gpurun_out/r12j_bench.err; echo "bench mini-E exit $?"
timeout 1500 python bench.py --workload configD_k101 --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > gpurun_out/r12j_bench_configD.json 2>> gpurun_out/r12j_bench.err; echo "bench D exit $?"
python - <<'PY'
import json
for f in ("r12j_bench_configC", "r12j_bench_configEprime", "r12j_bench_configEmini", "r12j_bench_configD"):
    try:
        j = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        r = j["roofline"]
        print(f, {k: j.get(k) for k in ("value", "ms_per_step", "ms_per_step_median", "ms_per_step_max", "launches_per_build", "host_round_trips_per_build")}, "hbm", j["hbm_resident"]["ms_per_step"],
              "roofline", {k: r.get(k) for k in ("kernel_ms", "traffic", "traffic_raw", "traffic_streaming_x2", "frac", "frac_range", "waste", "fetch_factor_applied")})
        for o in j.get("roofline_other", []): print("   ", o["kernel"][:40], {k: o.get(k) for k in ("kernel_ms", "traffic", "frac", "waste")})
        if j.get("cpu_baseline"): print("    cpu", {k: j["cpu_baseline"].get(k) for k in ("value", "cores", "host_threads_in_t_hot")}, (j["cpu_baseline"].get("recorded_whole_workload") or {}).get("value"))
        if j.get("t_e2e"): print("    e2e", j["t_e2e"].get("wall_s"), (j["t_e2e"].get("cli_fresh_process") or {}).get("wall_s"))
    except Exception as e:
        print(f, "failed", e)
PY
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/r12j_prof -o stats -- python $OLDPWD/bench.py --steps 3 --warmup 1 --init-builds 0 --no-cpu-baseline --no-e2e --pmc off > $OLDPWD/gpurun_out/r12j_prof_bench.json 2> $OLDPWD/gpurun_out/r12j_prof.err; echo "rocprof exit $?"; cd $OLDPWD
DB=$(find gpurun_out/r12j_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r12j_kernel_stats_configC.csv && head -14 gpurun_out/r12j_kernel_stats_configC.csv
find gpurun_out/r12j_prof -type f -size +8M -delete
export AC_NO_TORCH=1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/r12j_prof2 -o stats -- python $OLDPWD/tools/ab_knobs.py --steps 4 --variants "base" > $OLDPWD/gpurun_out/r12j_prof_ab.json 2> $OLDPWD/gpurun_out/r12j_prof2.err; cd $OLDPWD
DB=$(find gpurun_out/r12j_prof2 -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r12j_kernel_stats_driver_configC.csv
[ -n "$DB" ] && python tools/rocpd_launches.py $DB "" 5000 0 > gpurun_out/r12j_timeline_driver_configC.txt 2>/dev/null
find gpurun_out/r12j_prof2 -type f -size +8M -delete
tail -n 4 gpurun_out/r12j_bench.err
