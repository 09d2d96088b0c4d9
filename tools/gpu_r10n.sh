#!/bin/bash
# visit m: the upload pipeline taken apart (ring memory kinds, pack-then-copy, who waits for whom)
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 tools/microbench/upload_probe.hip -o /tmp/upload_probe -Lautocycler_amd -lautocycler_hip -Wl,-rpath,$PWD/autocycler_amd -pthread 2> gpurun_out/r10n_build.err || { tail -5 gpurun_out/r10n_build.err; exit 1; }
for NT in 0; do
  AC_PACK_NT=$NT timeout 300 /tmp/upload_probe > gpurun_out/r10n_upload_probe_nt$NT.jsonl 2> gpurun_out/r10n_probe.err; echo "probe nt=$NT exit $?"
  cat gpurun_out/r10n_upload_probe_nt$NT.jsonl
done
