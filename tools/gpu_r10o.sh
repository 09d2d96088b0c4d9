#!/bin/bash
# visit o: upload in smaller chunks over several copy queues (probe r10n: 3.2-3.4 -> 2.8-2.95 ms for config C's 122 MB)
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "| upload_device_ms", round(j.get("upload_device_ms", 0) or 0, 3), "insert_k", round(j.get("insert_kernel_ms", 0), 3), j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
V="base;AC_UPLOAD_CHUNK_MB=8,AC_UPLOAD_STREAMS=2;AC_UPLOAD_CHUNK_MB=8,AC_UPLOAD_STREAMS=3;AC_UPLOAD_CHUNK_MB=16,AC_UPLOAD_STREAMS=2;AC_UPLOAD_CHUNK_MB=16,AC_UPLOAD_STREAMS=3;AC_UPLOAD_CHUNK_MB=32,AC_UPLOAD_STREAMS=2;AC_UPLOAD_CHUNK_MB=8,AC_UPLOAD_STREAMS=3,AC_UPLOAD_THREADS=48;AC_UPLOAD_CHUNK_MB=8,AC_UPLOAD_STREAMS=3,AC_UPLOAD_PIECE_MB=64;AC_UPLOAD_CHUNK_MB=8,AC_UPLOAD_STREAMS=3,AC_UPLOAD_PIECE_MB=16;base;AC_UPLOAD_CHUNK_MB=8,AC_UPLOAD_STREAMS=3;AC_UPLOAD_CHUNK_MB=4,AC_UPLOAD_STREAMS=4"
timeout 600 python tools/ab_knobs.py --steps 10 --host-entry --variants "$V" > gpurun_out/r10o_ab_upload_chunks_streams_host_entry_configC.jsonl 2> gpurun_out/r10o.err; echo "ab exit $?"; show gpurun_out/r10o_ab_upload_chunks_streams_host_entry_configC.jsonl
tail -3 gpurun_out/r10o.err
