#!/bin/bash
# Instruction-issue picture of the big kernels: VALU / VMEM activity against wave cycles (one rocprofv3 --pmc pass per group).
# Usage: tools/pmc_sq.sh TAG
TAG=${1:-rXX}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
i=0
for GRP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY" "TCC_HIT TCC_MISS TCC_REQ TCC_ATOMIC"; do
  i=$((i+1))
  cd /tmp && timeout 600 rocprofv3 --pmc $GRP --kernel-trace -f csv -d $R/gpurun_out/${TAG}_sq$i -o pmc -- \
      python $R/bench.py --steps 1 --warmup 0 --init-builds 0 --init-seconds 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/${TAG}_sq$i.err
  cd $R
  F=$(find gpurun_out/${TAG}_sq$i -name '*counter_collection.csv' | head -1)
  python - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for key in ("DegreeFunctor", "insert_wave", "PathWalkFunctor", "ExpandFunctor", "RemapFunctor", "MarkFunctor"):
        if key in n:
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k, {a: f"{b:.3g}" for a, b in v.items()})
PY
  find gpurun_out/${TAG}_sq$i -type f -size +2M -delete
done
