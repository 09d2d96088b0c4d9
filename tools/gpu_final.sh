#!/bin/bash
# bench.py plain, then the same command under rocprofv3 --kernel-trace --stats (summary -> profiles/).  Usage: tools/gpu_final.sh TAG
TAG=${1:-rXX}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 400 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
cut -c1-400 gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o prof -- python $R/bench.py --steps 3 --warmup 1 --init-builds 0 --init-seconds 0 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err; echo "rocprof exit $?"
cd $R
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_kernel_stats.csv; head -8 gpurun_out/${TAG}_kernel_stats.csv; python tools/rocpd_launches.py $DB insert_wave 9 9 > gpurun_out/${TAG}_insert_launches.txt; rm -f $DB; fi
