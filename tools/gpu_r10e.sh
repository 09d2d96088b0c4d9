#!/bin/bash
# Round-4 visit e: the multi-device code on the device box (ranks sharing the one GPU; RCCL world of one), what the protocol moves now, the
# torch driver's new exchanges, and where a fresh CLI process spends its time.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multi_gpu.py tests/test_sharded_gpu.py tests/test_gpu_boundary.py -x -q > gpurun_out/r10e_pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r10e_pytest.log
export AC_NO_TORCH=1
timeout 400 python tools/multi_bench.py --steps 3 --worlds 1,2,4,8 > gpurun_out/r10e_multi_entry_one_device_configC.jsonl 2> gpurun_out/r10e_multi.err; echo "multi exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/r10e_multi_entry_one_device_configC.jsonl"):
    j = json.loads(l)
    m = j.get("multi") or {}
    print({k: j.get(k) for k in ("variant", "world", "ms_median", "ms", "gfa_md5") if k in j}, {k: m.get(k) for k in ("n_ranks", "bytes_fragments", "bytes_bitmap", "bytes_degrees", "bytes_links", "bytes_queries", "bytes_answers", "bytes_reduce", "candidates_total", "candidates_owned_max")})
PY
# fresh-process CLI on config C
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
from autocycler_amd import synth
t = time.time()
synth.write_fasta_dir(synth.WORKLOADS["configC_k51"][2](), "/dev/shm/ac_cli_in")
print("fasta written", round(time.time() - t, 1), "s")
PY
for i in 1 2 3; do
  rm -rf /dev/shm/ac_cli_out
  t0=$(date +%s.%N)
  AC_DEBUG_WARM=1 ./autocycler_amd/autocycler-compress compress -i /dev/shm/ac_cli_in -a /dev/shm/ac_cli_out --kmer 51 -t 32 2>&1 | tail -20 | tr '\n' ';'
  t1=$(date +%s.%N)
  echo " WALL $(echo "$t1 - $t0" | bc) s"
done > gpurun_out/r10e_cli_fresh_process.txt 2>&1
cat gpurun_out/r10e_cli_fresh_process.txt | cut -c1-1500
rm -rf /dev/shm/ac_cli_in /dev/shm/ac_cli_out
