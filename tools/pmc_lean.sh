#!/bin/bash
# HBM-side traffic per kernel from the L2 counters (FETCH_SIZE, WRITE_SIZE: one counter per pass, --kernel-trace only), on the
# torch-free driver tools/ab_knobs.py (7 builds of config C per pass: 2 + 2 untimed, 2 timed, 1 with stage timers).
# Usage: tools/pmc_lean.sh TAG [VARIANT] [WORKLOAD]      (VARIANT: an ab_knobs variant such as AC_PATH_COPY=1; default base;
#        WORKLOAD: a name of autocycler_amd.synth.WORKLOADS such as configEprime_k51; default: config C)
TAG=${1:-rXX}
VARIANT=${2:-base}
WORKLOAD=${3:-}
WL_ARGS=""
[ -n "$WORKLOAD" ] && WL_ARGS="--workload $WORKLOAD"
export TMPDIR=/tmp AC_NO_TORCH=1
R=$PWD
mkdir -p $R/gpurun_out
for CTR in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout ${PMC_TIMEOUT:-100} rocprofv3 --pmc $CTR --kernel-trace -f csv -d $R/gpurun_out/${TAG}_pmc_$CTR -o pmc -- \
      python $R/tools/ab_knobs.py --variants "$VARIANT" --steps 2 $WL_ARGS > $R/gpurun_out/${TAG}_pmc_${CTR}_run.jsonl 2> $R/gpurun_out/${TAG}_pmc_$CTR.err
  echo "$CTR pass exit $?"
  cd $R
  F=$(find gpurun_out/${TAG}_pmc_$CTR -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then python tools/pmc_summary.py $F 7 > gpurun_out/${TAG}_pmc_$CTR.csv; head -8 gpurun_out/${TAG}_pmc_$CTR.csv; else tail -5 gpurun_out/${TAG}_pmc_$CTR.err; fi
  find gpurun_out/${TAG}_pmc_$CTR -type f -size +4M -delete
done
