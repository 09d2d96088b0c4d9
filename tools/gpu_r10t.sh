#!/bin/bash
# visit t: code-object load time against file size (stripped / unstripped main and W = 2 objects)
mkdir -p gpurun_out
hipcc -O2 --offload-arch=gfx950 tools/microbench/load_probe.hip -o /tmp/load_probe 2> gpurun_out/r10t_build.err || { tail -3 gpurun_out/r10t_build.err; exit 1; }
/tmp/load_probe build/cobj/main.co build/cobj/main_stripped.co build/cobj/w2.co build/cobj/w2_stripped.co > gpurun_out/r10t_load_probe.jsonl; cat gpurun_out/r10t_load_probe.jsonl
