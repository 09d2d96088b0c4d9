#!/bin/bash
# Calibrates the L2 counters (FETCH_SIZE, WRITE_SIZE) on the access pattern the k-mer insert and the probing stages have — random 8-byte
# reads and random compare-and-swaps, one per lane, over tables of 128 MB / 1 GB / 8 GB (the x2 of the guide is
# calibrated on wide streaming reads).  The kernels are the library's own RbReadFunctor / RbClaimFunctor (ac_random_access_ceilings_at:
# 48 M reads, min(capacity / 2, 48 M) claims per repetition, 4 repetitions); one counter per pass, --kernel-trace only.
# Usage: tools/pmc_calibrate_random.sh TAG   ->  gpurun_out/${TAG}_pmc_calibration_random.json
TAG=${1:-rXX}
export TMPDIR=/tmp AC_NO_TORCH=1
R=$PWD
mkdir -p $R/gpurun_out
for CTR in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout ${PMC_TIMEOUT:-200} rocprofv3 --pmc $CTR --kernel-trace -f csv -d $R/gpurun_out/${TAG}_cal_$CTR -o pmc -- \
      python $R/tools/pmc_calibrate_random.py run > $R/gpurun_out/${TAG}_cal_${CTR}_run.jsonl 2> $R/gpurun_out/${TAG}_cal_$CTR.err
  echo "$CTR pass exit $?"
  cd $R
done
F=$(find gpurun_out/${TAG}_cal_FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/${TAG}_cal_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_calibrate_random.py summarise "$F" "$W" gpurun_out/${TAG}_cal_FETCH_SIZE_run.jsonl > gpurun_out/${TAG}_pmc_calibration_random.json && cat gpurun_out/${TAG}_pmc_calibration_random.json
find gpurun_out/${TAG}_cal_FETCH_SIZE gpurun_out/${TAG}_cal_WRITE_SIZE -type f -size +4M -delete
