#!/bin/bash
# ONE runner for the visits to the MI355X box (round 5: replaces the ~45 one-off tools/gpu_r1*.sh of rounds 3-4, which stay in the history).
#   gpurun --timeout 3000 -- 'bash tools/gpu_visit.sh TAG step [step ...]'
# Every step writes under gpurun_out/ with the TAG as prefix; what is to be kept is copied to profiles/ afterwards (README.md there).
# Steps:
#   smoke             __graft_entry__.smoke()
#   tests[:EXPR]      pytest -m gpu (optionally -k EXPR)                                   -> TAG_pytest_gpu.log
#   prims             the hand-written scan / radix sort against std:: + the verifier tests  -> TAG_pytest_prims.log
#   bench[:WORKLOAD]  bench.py (config C: 20 steps, e2e, cpu baseline; a named workload: 5-10 steps, no e2e)  -> TAG_bench_<workload>.json
#   protocol[:WL]     bench.py --mode sharded --protocol-always at N = 1 beside the direct dispatch           -> TAG_bench_protocol_n1_*.json
#   ab[:WORKLOAD]     tools/ab_knobs.py base;base or $AB_VARIANTS (torch-free driver: stage table, launches, round trips, md5) -> TAG_ab_<workload>.jsonl
#   calibrate         FETCH_SIZE / WRITE_SIZE per random access (tools/pmc_calibrate_random.sh)                -> TAG_pmc_calibration_random.json
#   pmc[:WORKLOAD]    the two PMC passes + pmc_traffic json stamped with the source hash                       -> pmc_traffic[_WORKLOAD].json
#   prof              rocprofv3 --kernel-trace --stats of bench.py (config C) and of the torch-free driver     -> TAG_kernel_stats_*.csv, TAG_timeline_driver_configC.txt
#   multi             ac_compress_build_multi with 1 / 2 / 4 / 8 ranks sharing the device (bytes per exchange) -> TAG_multi_entry_one_device_configC.jsonl
#   multijob          the same entry, the 8-species bench job over 8 ranks sharing the device                   -> TAG_multi_entry_one_device_benchjob8.jsonl
#   cli               three fresh autocycler-compress processes on config C with the warm-up breakdown          -> TAG_cli_fresh_process.txt
#   superkmer         tools/microbench/superkmer_bench.hip on E' and config B                                   -> TAG_superkmer_bench_*.jsonl
export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out
R=$PWD
summ() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], {k: j.get(k) for k in ("value", "ms_per_step", "ms_per_step_median", "ms_per_step_max", "launches_per_build", "host_round_trips_per_build")},
          "hbm", (j.get("hbm_resident") or {}).get("ms_per_step"), "roofline", {k: r.get(k) for k in ("kernel_ms", "traffic", "frac", "frac_range", "waste", "traffic_source")})
    st = j.get("stages_s") or {}
    print("   stages ms", {k: round(v * 1e3, 3) for k, v in st.items()})
    if j.get("cpu_baseline"): print("   cpu", {k: j["cpu_baseline"].get(k) for k in ("value", "cores", "host_threads_in_t_hot")})
    if j.get("t_e2e"): print("   e2e warm", j["t_e2e"].get("wall_s"), "fresh CLI", (j["t_e2e"].get("cli_fresh_process") or {}).get("wall_s"))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
for STEP in "$@"; do
  NAME=${STEP%%:*}; ARG=""; [ "$STEP" != "$NAME" ] && ARG=${STEP#*:}
  case $NAME in
    smoke) python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ;;
    tests) if [ -n "$ARG" ]; then timeout 2400 python -m pytest tests -m gpu -x -q -k "$ARG" > gpurun_out/${TAG}_pytest_gpu.log 2>&1; else timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; fi
           echo "pytest exit $?"; tail -4 gpurun_out/${TAG}_pytest_gpu.log ;;
    prims) timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -k "primitives or verify" > gpurun_out/${TAG}_pytest_prims.log 2>&1; echo "prims exit $?"; tail -3 gpurun_out/${TAG}_pytest_prims.log ;;
    bench) if [ -z "$ARG" ]; then timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_configC.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"; summ gpurun_out/${TAG}_bench_configC.json
           else timeout 1500 python bench.py --workload $ARG --steps 8 --warmup 3 --no-e2e > gpurun_out/${TAG}_bench_${ARG}.json 2>> gpurun_out/${TAG}_bench.err; echo "bench $ARG exit $?"; summ gpurun_out/${TAG}_bench_${ARG}.json; fi ;;
    protocol) WL=""; [ -n "$ARG" ] && WL="--workload $ARG"
           timeout 600 python bench.py $WL --mode sharded --protocol-always --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --pmc off > gpurun_out/${TAG}_bench_protocol_n1_${ARG:-configC}.json 2>> gpurun_out/${TAG}_bench.err; summ gpurun_out/${TAG}_bench_protocol_n1_${ARG:-configC}.json
           timeout 600 python bench.py $WL --mode sharded --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --pmc off > gpurun_out/${TAG}_bench_sharded_direct_n1_${ARG:-configC}.json 2>> gpurun_out/${TAG}_bench.err; summ gpurun_out/${TAG}_bench_sharded_direct_n1_${ARG:-configC}.json ;;
    ab)    WL=""; [ -n "$ARG" ] && WL="--workload $ARG"
           AC_NO_TORCH=1 timeout 600 python tools/ab_knobs.py $WL --steps 12 --variants "${AB_VARIANTS:-base;base}" > gpurun_out/${TAG}_ab_${ARG:-configC}.jsonl 2>> gpurun_out/${TAG}.err
           python - gpurun_out/${TAG}_ab_${ARG:-configC}.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j: print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "launches", j.get("launches"), "round trips", j.get("readbacks"), j.get("stages_ms"), j.get("expand"), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
           ;;
    calibrate) bash tools/pmc_calibrate_random.sh $TAG > gpurun_out/${TAG}_cal.log 2>&1; tail -6 gpurun_out/${TAG}_cal.log; cp gpurun_out/${TAG}_pmc_calibration_random.json profiles/pmc_calibration_random.json ;;
    pmc)   WL=${ARG:-configC_k51}; S=$TAG; [ -n "$ARG" ] && S=${TAG}_$ARG
           PMC_TIMEOUT=900 bash tools/pmc_lean.sh $S base $ARG > gpurun_out/${S}_pmc.log 2>&1
           OUT=pmc_traffic.json; [ -n "$ARG" ] && OUT=pmc_traffic_$ARG.json
           python tools/pmc_traffic.py gpurun_out/${S}_pmc_FETCH_SIZE.csv gpurun_out/${S}_pmc_WRITE_SIZE.csv $(python tools/workload_n_text.py $WL) $S 7 $WL > gpurun_out/$OUT && cp gpurun_out/$OUT profiles/$OUT
           python -c "import json; d = json.load(open('gpurun_out/$OUT')); print('$OUT', d['source_hash'], 'raw', round(d['traffic_raw'] / 1e9, 3), 'GB, x2', round(d['traffic_streaming_x2'] / 1e9, 3), 'applied', round(d['traffic_bytes_per_build'] / 1e9, 3), 'factor', d['fetch_correction'])" ;;
    prof)  cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o stats -- python $R/bench.py --steps 3 --warmup 1 --init-builds 0 --no-cpu-baseline --no-e2e --pmc off > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof.err; echo "rocprof (bench.py) exit $?"; cd $R
           DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_kernel_stats_configC.csv && head -12 gpurun_out/${TAG}_kernel_stats_configC.csv
           find gpurun_out/${TAG}_prof -type f -size +8M -delete
           cd /tmp && AC_NO_TORCH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof2 -o stats -- python $R/tools/ab_knobs.py --steps 4 --variants "base" > $R/gpurun_out/${TAG}_prof_ab.json 2> $R/gpurun_out/${TAG}_prof2.err; cd $R
           DB=$(find gpurun_out/${TAG}_prof2 -name '*.db' | head -1)
           [ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_kernel_stats_driver_configC.csv && python tools/rocpd_launches.py $DB "" 5000 0 > gpurun_out/${TAG}_timeline_driver_configC.txt 2>/dev/null
           find gpurun_out/${TAG}_prof2 -type f -size +8M -delete ;;
    multi) AC_NO_TORCH=1 timeout 600 python tools/multi_bench.py --steps 3 --worlds 1,2,4,8 > gpurun_out/${TAG}_multi_entry_one_device_configC.jsonl 2> gpurun_out/${TAG}_multi.err; echo "multi exit $?"
           python - gpurun_out/${TAG}_multi_entry_one_device_configC.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l); m = j.get("multi") or {}
    print({k: j.get(k) for k in ("variant", "ms_median", "gfa_md5") if k in j}, {k: v for k, v in m.items() if k.startswith("bytes_") or k in ("n_ranks", "transport", "degrees_open", "candidates_owned_max")})
PY
           ;;
    multijob) AC_NO_TORCH=1 timeout 900 python tools/multi_bench.py --workload benchjob8_k51 --steps 2 --warmup 1 --worlds 8 > gpurun_out/${TAG}_multi_entry_one_device_benchjob8.jsonl 2>> gpurun_out/${TAG}_multi.err; echo "multijob exit $?"
           python - gpurun_out/${TAG}_multi_entry_one_device_benchjob8.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l); m = j.get("multi") or {}
    print(j.get("variant"), "| ms", round(j.get("ms_median", 0), 1), "| received by one rank at most", m.get("bytes_received_max"), "| md5", j.get("gfa_md5", "")[:8])
PY
           ;;
    cli)   python -c "
import sys, time
sys.path.insert(0, '.')
from autocycler_amd import synth
synth.write_fasta_dir(synth.WORKLOADS['configC_k51'][2](), '/dev/shm/ac_cli_in')"
           for i in 1 2 3; do rm -rf /dev/shm/ac_cli_out; t0=$(date +%s.%N)
             AC_DEBUG_WARM=1 ./autocycler_amd/autocycler-compress compress -i /dev/shm/ac_cli_in -a /dev/shm/ac_cli_out --kmer 51 -t 32 2>&1 | tail -20 | tr '\n' ';'
             t1=$(date +%s.%N); echo " WALL $(python -c "print(round($t1 - $t0, 3))") s"; done > gpurun_out/${TAG}_cli_fresh_process.txt 2>&1
           cut -c1-900 gpurun_out/${TAG}_cli_fresh_process.txt; rm -rf /dev/shm/ac_cli_in /dev/shm/ac_cli_out ;;
    superkmer) hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench/superkmer_bench.hip -o /tmp/superkmer_bench 2> gpurun_out/${TAG}_build.err || { tail -3 gpurun_out/${TAG}_build.err; continue; }
           python -c "
import sys
import numpy as np
sys.path.insert(0, '.')
from autocycler_amd import synth
for name in ('configEprime_k51', 'configB_k51'):
    parts = []
    for contigs in synth.WORKLOADS[name][2]():
        for _, s in contigs:
            parts.append(np.asarray(s, dtype=np.uint8)); parts.append(np.frombuffer(b'\$', dtype=np.uint8))
    np.concatenate(parts).tofile(f'/tmp/{name}.bin')"
           for WL in configEprime_k51 configB_k51; do : > gpurun_out/${TAG}_superkmer_bench_$WL.jsonl
             for M in 21 25; do for B in 13 14; do timeout 120 /tmp/superkmer_bench /tmp/$WL.bin 51 $M $B >> gpurun_out/${TAG}_superkmer_bench_$WL.jsonl 2>> gpurun_out/${TAG}.err; done; done
             cat gpurun_out/${TAG}_superkmer_bench_$WL.jsonl | cut -c1-400; done ;;
    *) echo "unknown step $STEP" ;;
  esac
done
