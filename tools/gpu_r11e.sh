#!/bin/bash
# Round-4 closing visit: the device suite, stamped PMC traffic for config C and E', the bench lines, kernel statistics, smoke.
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r11e_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r11e_pytest_gpu.log
# stamped traffic files (what bench.py reads when the library's source hash matches)
PMC_TIMEOUT=200 bash tools/pmc_lean.sh r11e base > gpurun_out/r11e_pmc_C.log 2>&1
PMC_TIMEOUT=300 bash tools/pmc_lean.sh r11eE base configEprime_k51 > gpurun_out/r11e_pmc_E.log 2>&1
python tools/pmc_traffic.py gpurun_out/r11e_pmc_FETCH_SIZE.csv gpurun_out/r11e_pmc_WRITE_SIZE.csv 487508684 r11e 7 configC_k51 > gpurun_out/pmc_traffic.json && cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
python tools/pmc_traffic.py gpurun_out/r11eE_pmc_FETCH_SIZE.csv gpurun_out/r11eE_pmc_WRITE_SIZE.csv 101448839 r11eE 7 configEprime_k51 > gpurun_out/pmc_traffic_configEprime_k51.json && cp gpurun_out/pmc_traffic_configEprime_k51.json profiles/pmc_traffic_configEprime_k51.json
python -c "
import json
for f in ('gpurun_out/pmc_traffic.json', 'gpurun_out/pmc_traffic_configEprime_k51.json'):
    d = json.load(open(f)); print(f, d['source_hash'], round(d['traffic_bytes_per_build'] / 1e9, 3), 'GB insert', d['calibration']['fetch_raw_over_known'])
"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r11e_bench_configC.json 2> gpurun_out/r11e_bench.err; echo "bench exit $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r11e_bench_configC.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "steps", "warmup")}, "hbm_resident", j["hbm_resident"]["ms_per_step"], "roofline", {k: j["roofline"].get(k) for k in ("kernel_ms", "traffic", "frac", "traffic_source", "traffic_note")})
print("t_e2e", {k: j["t_e2e"].get(k) for k in ("wall_s",)}, "cli", (j["t_e2e"].get("cli_fresh_process") or {}).get("wall_s"), "cpu", j["cpu_baseline"]["value"])
PY
timeout 900 python bench.py --workload configEprime_k51 --steps 10 --warmup 3 --no-e2e > gpurun_out/r11e_bench_configEprime.json 2> gpurun_out/r11e_benchE.err; echo "bench E' exit $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r11e_bench_configEprime.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step")}, "hbm_resident", j["hbm_resident"]["ms_per_step"], "roofline", {k: j["roofline"].get(k) for k in ("kernel_ms", "traffic", "frac", "waste", "traffic_note")})
for o in j["roofline_other"]: print(o["kernel"][:40], {k: o.get(k) for k in ("kernel_ms", "traffic", "frac", "waste")})
PY
# kernel statistics of the bench command
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/r11e_prof -o stats -- python $OLDPWD/bench.py --steps 3 --warmup 1 --init-builds 0 --no-cpu-baseline --no-e2e --pmc off > $OLDPWD/gpurun_out/r11e_prof_bench.json 2> $OLDPWD/gpurun_out/r11e_prof.err; echo "rocprof exit $?"; cd $OLDPWD
DB=$(find gpurun_out/r11e_prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r11e_kernel_stats_configC.csv && head -12 gpurun_out/r11e_kernel_stats_configC.csv
[ -n "$DB" ] && python tools/rocpd_launches.py $DB "" 400 0 > gpurun_out/r11e_timeline_all.txt 2>/dev/null
find gpurun_out/r11e_prof -type f -size +8M -delete
