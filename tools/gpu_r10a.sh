#!/bin/bash
# Round-4 opening visit: the diverse regime (E') on the code round 3 closed with — stage table, kernel stats + timeline, PMC traffic.
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
timeout 200 python tools/ab_knobs.py --workload configEprime_k51 --steps 8 --variants "base;base" > gpurun_out/r10a_ab_configEprime_k51.jsonl 2> gpurun_out/r10a_ab.err; echo "ab exit $?"
cat gpurun_out/r10a_ab_configEprime_k51.jsonl | cut -c1-900
timeout 250 tools/gpu_timeline.sh r10a_configEprime --workload configEprime_k51
head -30 gpurun_out/r10a_configEprime_kernel_stats.csv
PMC_TIMEOUT=150 timeout 320 tools/pmc_lean.sh r10a_configEprime base configEprime_k51
