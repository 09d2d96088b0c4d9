#!/bin/bash
# Round-5 opening visit: today's baseline on this box — config C and E' through the single-device and the sharded (world 1) drivers.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --pmc file > gpurun_out/r12a_bench_configC.json 2> gpurun_out/r12a_bench.err; echo "bench exit $?"
timeout 300 python bench.py --mode sharded --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --pmc off > gpurun_out/r12a_bench_sharded_n1_configC.json 2>> gpurun_out/r12a_bench.err; echo "sharded exit $?"
timeout 400 python bench.py --workload configEprime_k51 --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --pmc file > gpurun_out/r12a_bench_configEprime.json 2>> gpurun_out/r12a_bench.err; echo "E' exit $?"
python - <<'PY'
import json
for f in ("r12a_bench_configC", "r12a_bench_sharded_n1_configC", "r12a_bench_configEprime"):
    try:
        j = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: j.get(k) for k in ("value", "ms_per_step")}, "hbm", (j.get("hbm_resident") or {}).get("ms_per_step"), "stages", j.get("stages_s"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -5 gpurun_out/r12a_bench.err
