#!/bin/bash
# Round-5 visit c: the device verifier (ac_verify_graph) and the compact protocol exchanges on the device.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_boundary.py tests/test_multi_gpu.py tests/test_sharded_gpu.py -x -q > gpurun_out/r12c_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r12c_pytest.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "config_b or config_c_96 or fragmented or d_full" > gpurun_out/r12c_pytest_fullsize.log 2>&1; echo "fullsize exit $?"; grep -a "ac_verify_graph_device\|passed\|failed\|Error" gpurun_out/r12c_pytest_fullsize.log | tail -8
timeout 300 python bench.py --mode sharded --protocol-always --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --pmc off > gpurun_out/r12c_bench_protocol_n1_configC.json 2>> gpurun_out/r12c_bench.err; echo "protocol C exit $?"
export AC_NO_TORCH=1
timeout 400 python tools/multi_bench.py --steps 3 --worlds 1,2,4,8 > gpurun_out/r12c_multi_entry_one_device_configC.jsonl 2> gpurun_out/r12c_multi.err; echo "multi exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/r12c_multi_entry_one_device_configC.jsonl"):
    j = json.loads(l)
    m = j.get("multi") or {}
    print({k: j.get(k) for k in ("variant", "ms_median", "gfa_md5") if k in j}, {k: m.get(k) for k in ("n_ranks", "transport", "bytes_fragments", "bytes_bitmap", "bytes_sibling", "bytes_degrees", "degrees_open", "bytes_links", "bytes_queries", "bytes_answers", "bytes_reduce", "bytes_tail", "bytes_received_max")})
PY
tail -n 3 gpurun_out/r12c_bench.err gpurun_out/r12c_multi.err
