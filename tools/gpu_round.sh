#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Usage: tools/gpu_round.sh TAG [pytest|nopytest]
TAG=${1:-rXX}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${2:-pytest}" = "pytest" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest.log
  tail -3 gpurun_out/${TAG}_pytest.log
fi
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench exit $?"
tail -c 3000 gpurun_out/${TAG}_bench.json
R=$PWD
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o prof -- python $R/bench.py --steps 3 --warmup 1 --init-builds 0 --init-seconds 0 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err; echo "rocprof exit $?"
cd $R
ls -R gpurun_out/${TAG}_prof | head -20
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_kernel_stats.csv; head -25 gpurun_out/${TAG}_kernel_stats.csv; rm -f $DB; fi
CSV=$(find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -1)
if [ -n "$CSV" ]; then cp $CSV gpurun_out/${TAG}_kernel_stats_rocprof.csv; head -25 $CSV; fi
