#!/bin/bash
# Full-size golden results from the ORACLE (the C++ restatement of the reference CPU path, oracle/): writes the FASTA files of a
# synthetic configuration with the committed generator, runs `autocycler_oracle compress` on them and records the md5 of the GFA,
# the printed statistics and the stage times in tests/golden/NAME.json.  The device test
# tests/test_gpu_fullsize.py::test_gfa_digest_equals_the_oracle compares the digest of the GFA built on the MI355X from the same
# inputs with it.
#   bash tests/golden/make_configC_golden.sh                                  -> configC_k51.json       (BASELINE configs[2]: 96 x ~5 Mbp,
#                                                                                 k = 51; about 26 minutes and 16 GB on 8 vCPUs)
#   bash tests/golden/make_configC_golden.sh configDprime_k101 24 10000000 101 -> configDprime_k101.json (scaled replica of configs[3]:
#                                                                                 24 x ~10 Mbp, k = 101; about 12 minutes and 8 GB)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=${1:-configC_k51}; NASM=${2:-96}; GENOME=${3:-5000000}; K=${4:-51}
W=${GOLDEN_WORKDIR:-/tmp/golden_$NAME}
mkdir -p $W
make -C $ROOT/oracle > /dev/null
python - <<EOF
import sys
sys.path.insert(0, "$ROOT")
from autocycler_amd import synth
synth.write_fasta_dir(synth.make_assemblies($NASM, genome=$GENOME, plasmid=100_000, sub=1e-4, indel=1e-5, seed=51_000), "$W/in")
EOF
S=$(date +%s)
$ROOT/oracle/autocycler_oracle compress -i $W/in -a $W/out --kmer $K -t 8 2> $W/oracle.log
E=$(( $(date +%s) - S ))
python - <<EOF
import hashlib, json, re
log = open("$W/oracle.log").read()
m = re.search(r"times: load\+repair ([\d.]+) kmer_graph ([\d.]+) unitig_graph ([\d.]+) simplify ([\d.]+) save ([\d.]+)", log)
st = re.findall(r"(\d+) unitigs, (\d+) links\ntotal length: (\d+) bp", log)
data = open("$W/out/input_assemblies.gfa", "rb").read()
json.dump({"what": "autocycler_oracle compress (oracle/: C++ restatement of the reference CPU path)",
           "inputs": "synth.make_assemblies($NASM, genome=$GENOME, plasmid=100_000, sub=1e-4, indel=1e-5, seed=51_000) written as FASTA",
           "k": $K, "gfa_md5": hashlib.md5(data).hexdigest(), "gfa_bytes": len(data),
           "kmers": int(re.search(r"Graph contains (\d+) k-mers", log).group(1)),
           "pre": dict(zip(("unitigs", "links", "total_length"), map(int, st[0]))),
           "post": dict(zip(("unitigs", "links", "total_length"), map(int, st[1]))),
           "seconds": dict(zip(("load_and_end_repair_8_threads", "kmer_graph", "unitig_graph", "simplify", "save"), map(float, m.groups()))),
           "wall_seconds": $E}, open("$ROOT/tests/golden/$NAME.json", "w"), indent=1)
print(open("$ROOT/tests/golden/$NAME.json").read())
EOF
