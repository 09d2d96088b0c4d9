#!/bin/bash
# Round-5 visit i: the super-k-mer partitioned insert priced on E''s (and config B's) real text (tools/microbench/superkmer_bench.hip)
export TMPDIR=/tmp AC_NO_TORCH=1
mkdir -p gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/microbench/superkmer_bench.hip -o /tmp/superkmer_bench 2> gpurun_out/r12i_build.err || { tail -3 gpurun_out/r12i_build.err; exit 1; }
python - <<'PY'
import sys
import numpy as np
sys.path.insert(0, ".")
from autocycler_amd import synth
for name in ("configEprime_k51", "configB_k51"):
    asm = synth.WORKLOADS[name][2]()
    parts = []
    for contigs in asm:
        for _, s in contigs:
            parts.append(np.asarray(s, dtype=np.uint8)); parts.append(np.frombuffer(b"$", dtype=np.uint8))
    np.concatenate(parts).tofile(f"/tmp/{name}.bin")
    print(name, sum(len(p) for p in parts))
PY
for WL in configEprime_k51 configB_k51; do
  : > gpurun_out/r12i_superkmer_bench_$WL.jsonl
  for M in 21 25; do for B in 13 14; do
    timeout 120 /tmp/superkmer_bench /tmp/$WL.bin 51 $M $B >> gpurun_out/r12i_superkmer_bench_$WL.jsonl 2>> gpurun_out/r12i.err
  done; done
  python - $WL <<'PY'
import json, sys
for l in open(f"gpurun_out/r12i_superkmer_bench_{sys.argv[1]}.jsonl"):
    j = json.loads(l)
    print(sys.argv[1], {k: j.get(k) for k in ("m", "buckets", "records", "kmers_per_record", "bytes_per_kmer", "records_in_fullest_bucket", "mean_records_per_bucket", "count_ms", "scatter_ms", "novel_clear_ms", "dedup_ms", "sum_ms", "distinct_kmers", "kmers_inserted", "buckets_overflowed", "error")})
PY
done
tail -n 3 gpurun_out/r12i.err
