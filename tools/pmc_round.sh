#!/bin/bash
# HBM traffic of every kernel from the L2 memory-side counters, one counter per pass (FETCH_SIZE takes 3 of the
# 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Counter passes carry --kernel-trace only.
# Usage: tools/pmc_round.sh TAG [bench args...]
TAG=${1:-rXX}; shift
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
for CTR in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --pmc $CTR --kernel-trace -f csv -d $R/gpurun_out/${TAG}_pmc_$CTR -o pmc -- \
      python $R/bench.py --steps 2 --warmup 1 --init-builds 0 --init-seconds 0 --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_pmc_${CTR}_bench.json 2> $R/gpurun_out/${TAG}_pmc_$CTR.err
  echo "$CTR pass exit $?"
  cd $R
  F=$(find gpurun_out/${TAG}_pmc_$CTR -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then python tools/pmc_summary.py $F 5 > gpurun_out/${TAG}_pmc_$CTR.csv; head -12 gpurun_out/${TAG}_pmc_$CTR.csv; else tail -5 gpurun_out/${TAG}_pmc_$CTR.err; fi
  find gpurun_out/${TAG}_pmc_$CTR -type f -size +4M -delete
done
