#!/bin/bash
# visit y: the claim rate a locality-preserving table placement could be built on (tools/microbench/locality_cas.hip)
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/microbench/locality_cas.hip -o /tmp/locality_cas 2> gpurun_out/r10y_build.err || { tail -3 gpurun_out/r10y_build.err; exit 1; }
timeout 120 /tmp/locality_cas > gpurun_out/r10y_locality_cas.jsonl; cat gpurun_out/r10y_locality_cas.jsonl
