#!/bin/bash
# Extra measurements of one GPU-box visit: the sharded path at world_size 1, and the replicated (graph) part of an
# 8-rank weak-scaling job measured as a single-device 768-assembly build.  Usage: tools/gpu_extra.sh TAG
TAG=${1:-rXX}
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --mode sharded --no-cpu-baseline > gpurun_out/${TAG}_bench_sharded_w1.json 2> gpurun_out/${TAG}_bench_sharded_w1.err; echo "sharded w1 exit $?"
tail -c 2500 gpurun_out/${TAG}_bench_sharded_w1.json; tail -5 gpurun_out/${TAG}_bench_sharded_w1.err
timeout 900 python bench.py --steps 2 --warmup 1 --assemblies ${2:-384} --no-cpu-baseline > gpurun_out/${TAG}_bench_big.json 2> gpurun_out/${TAG}_bench_big.err; echo "big exit $?"
tail -c 2500 gpurun_out/${TAG}_bench_big.json; tail -5 gpurun_out/${TAG}_bench_big.err
