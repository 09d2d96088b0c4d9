#!/bin/bash
# The N > 1 code path of bench.py with every rank on ONE GPU (gloo moves the collectives through the host): not a scaling
# measurement — the ranks time-slice one device — but elapsed / N bounds the work ONE rank does in an N-rank job, and the stage
# tables show which part of it grows with N.  Usage: tools/shared_gpu_scaling.sh TAG "2 4 8"
TAG=${1:-rXX}; NS=${2:-"2 4"}
mkdir -p gpurun_out
for N in $NS; do
  BENCH_FORCE_DEVICE=0 BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps 3 --warmup 1 --init-builds 1 --init-seconds 0 --no-independent > gpurun_out/${TAG}_shared_gpu_n$N.json 2> gpurun_out/${TAG}_shared_gpu_n$N.err
  echo "N=$N exit $?"
  python - <<PY
import json
try:
    j = json.loads([l for l in open("gpurun_out/${TAG}_shared_gpu_n$N.json") if l.startswith("{")][-1])
    st = {k: round(v * 1e3, 2) for k, v in j["stages_s"].items()}
    print("N=$N ms_per_step", round(j["ms_per_step"], 2), "per rank <=", round(j["ms_per_step"] / $N, 2), "comm_s", j["sharded"].get("comm_s"), "unitigs", j["sharded"]["unitigs"], "distinct", j["sharded"]["distinct"])
    print("   rank-0 stages (ms):", st)
except Exception as e:
    print("parse failed", e); print(open("gpurun_out/${TAG}_shared_gpu_n$N.err").read()[-1500:])
PY
done
