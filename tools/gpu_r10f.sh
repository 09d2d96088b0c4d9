#!/bin/bash
# Round-4 visit f: north_star's partitioned insert priced piece by piece (tools/microbench/partition_bench.hip), the fresh CLI process,
# the bench line of config C (traffic counted live if the committed file is stale) and of E', kernel statistics.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 120 build/partition_bench > gpurun_out/r10f_partition_microbench.json 2> gpurun_out/r10f_partition.err; echo "partition_bench exit $?"; cat gpurun_out/r10f_partition_microbench.json
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
from autocycler_amd import synth
synth.write_fasta_dir(synth.WORKLOADS["configC_k51"][2](), "/dev/shm/ac_cli_in")
PY
for i in 1 2 3 4; do
  rm -rf /dev/shm/ac_cli_out
  python - <<'PY'
import subprocess, time
t = time.perf_counter()
p = subprocess.run(["./autocycler_amd/autocycler-compress", "compress", "-i", "/dev/shm/ac_cli_in", "-a", "/dev/shm/ac_cli_out", "--kmer", "51", "-t", "32"], capture_output=True, text=True, env={"AC_DEBUG_WARM": "1", "PATH": "/usr/bin:/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
w = time.perf_counter() - t
print("WALL %.3f s |" % w, " ; ".join(l.strip() for l in p.stderr.splitlines() if "warm" in l or "Stage" in l or "Time to run" in l))
PY
done > gpurun_out/r10f_cli_fresh_process.txt 2>&1
cat gpurun_out/r10f_cli_fresh_process.txt | cut -c1-700
rm -rf /dev/shm/ac_cli_in /dev/shm/ac_cli_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r10f_bench_configC.json 2> gpurun_out/r10f_bench.err; echo "bench exit $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r10f_bench_configC.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "steps", "warmup")}, "hbm_resident", j["hbm_resident"]["ms_per_step"], "roofline", {k: j["roofline"].get(k) for k in ("kernel_ms", "traffic", "frac", "traffic_source", "traffic_note")})
print("t_e2e", {k: j["t_e2e"].get(k) for k in ("wall_s",)}, "cli", (j["t_e2e"].get("cli_fresh_process") or {}).get("wall_s"), "cpu", j["cpu_baseline"]["value"])
PY
cp gpurun_out/bench_live_pmc_FETCH_SIZE.csv gpurun_out/r10f_pmc_FETCH_SIZE_configC.csv 2>/dev/null; cp gpurun_out/bench_live_pmc_WRITE_SIZE.csv gpurun_out/r10f_pmc_WRITE_SIZE_configC.csv 2>/dev/null
timeout 900 python bench.py --workload configEprime_k51 --steps 10 --warmup 3 --no-e2e > gpurun_out/r10f_bench_configEprime.json 2> gpurun_out/r10f_benchE.err; echo "bench E' exit $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r10f_bench_configEprime.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step")}, "hbm_resident", j["hbm_resident"]["ms_per_step"], "roofline", {k: j["roofline"].get(k) for k in ("kernel_ms", "traffic", "frac", "waste", "traffic_note")})
for o in j["roofline_other"]: print(o["kernel"][:40], {k: o.get(k) for k in ("kernel_ms", "traffic", "frac", "waste")})
PY
cp gpurun_out/bench_live_pmc_FETCH_SIZE.csv gpurun_out/r10f_pmc_FETCH_SIZE_configEprime.csv 2>/dev/null; cp gpurun_out/bench_live_pmc_WRITE_SIZE.csv gpurun_out/r10f_pmc_WRITE_SIZE_configEprime.csv 2>/dev/null
