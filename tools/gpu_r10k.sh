#!/bin/bash
# visit k: what bounds the T_hot upload — host packers, link, or the pipelining (tools/microbench/upload_probe.hip)
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/microbench/upload_probe.hip -o /tmp/upload_probe -Lautocycler_amd -lautocycler_hip -Wl,-rpath,$PWD/autocycler_amd -pthread 2> gpurun_out/r10k_build.err || { tail -5 gpurun_out/r10k_build.err; exit 1; }
lscpu | grep -i "model name\|socket\|numa\|thread\|core" > gpurun_out/r10k_lscpu.txt
cat /sys/kernel/mm/transparent_hugepage/enabled >> gpurun_out/r10k_lscpu.txt
timeout 300 /tmp/upload_probe > gpurun_out/r10k_upload_probe.jsonl 2> gpurun_out/r10k_probe.err
echo "probe exit $?"
cat gpurun_out/r10k_lscpu.txt
cat gpurun_out/r10k_upload_probe.jsonl
