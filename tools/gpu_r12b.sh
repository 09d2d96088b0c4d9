#!/bin/bash
# Round-5 visit b: the protocol after the sibling bits / probe-free degrees / host-side path numbers / 40-byte link exchange:
# device parity of both drivers, the protocol at world 1 (every phase, no bytes moved) against the single-device build, ranks sharing the device.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_multi_gpu.py tests/test_sharded_gpu.py -x -q > gpurun_out/r12b_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r12b_pytest.log
for i in 1 2; do
timeout 300 python bench.py --mode sharded --protocol-always --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --pmc off > gpurun_out/r12b_bench_protocol_n1_configC_$i.json 2>> gpurun_out/r12b_bench.err; echo "protocol C exit $?"
timeout 300 python bench.py --mode sharded --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --pmc off > gpurun_out/r12b_bench_sharded_direct_n1_configC_$i.json 2>> gpurun_out/r12b_bench.err; echo "direct C exit $?"
done
timeout 400 python bench.py --workload configEprime_k51 --mode sharded --protocol-always --steps 8 --warmup 2 --no-e2e --no-cpu-baseline --pmc off > gpurun_out/r12b_bench_protocol_n1_configEprime.json 2>> gpurun_out/r12b_bench.err; echo "protocol E' exit $?"
timeout 400 python bench.py --workload configEprime_k51 --steps 8 --warmup 2 --no-e2e --no-cpu-baseline --no-host-bracket --pmc off > gpurun_out/r12b_bench_single_configEprime.json 2>> gpurun_out/r12b_bench.err; echo "single E' exit $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r12b_bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        st = j.get("stages_s") or {}
        print(f.split("r12b_bench_")[1], {k: round(j.get(k), 3) for k in ("ms_per_step",)}, "hbm", (j.get("hbm_resident") or {}).get("ms_per_step"), {k: round(v * 1e3, 3) for k, v in st.items()})
    except Exception as e:
        print(f, "failed", e)
PY
export AC_NO_TORCH=1
timeout 400 python tools/multi_bench.py --steps 3 --worlds 1,2,4,8 > gpurun_out/r12b_multi_entry_one_device_configC.jsonl 2> gpurun_out/r12b_multi.err; echo "multi exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/r12b_multi_entry_one_device_configC.jsonl"):
    j = json.loads(l)
    m = j.get("multi") or {}
    print({k: j.get(k) for k in ("variant", "world", "ms_median", "ms", "gfa_md5") if k in j}, {k: m.get(k) for k in ("n_ranks", "transport", "bytes_fragments", "bytes_bitmap", "bytes_degrees", "bytes_links", "bytes_queries", "bytes_answers", "bytes_reduce", "candidates_total", "candidates_owned_max")})
PY
tail -3 gpurun_out/r12b_bench.err gpurun_out/r12b_multi.err
