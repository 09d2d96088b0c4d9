#!/bin/bash
# visit: host threads packing straight into device memory through the PCIe BAR (tools/microbench/bar_write_probe.hip)
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/microbench/bar_write_probe.hip -o /tmp/bar_write_probe -Lautocycler_amd -lautocycler_hip -Wl,-rpath,$PWD/autocycler_amd -pthread 2> gpurun_out/r11a_build.err || { tail -3 gpurun_out/r11a_build.err; exit 1; }
timeout 200 /tmp/bar_write_probe > gpurun_out/r11a_bar_write_probe.jsonl 2> gpurun_out/r11a.err; echo "exit $?"; cat gpurun_out/r11a_bar_write_probe.jsonl; tail -3 gpurun_out/r11a.err
