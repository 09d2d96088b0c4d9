#!/bin/bash
# Full-size golden result for BASELINE.json configs[2] (96 x ~5 Mbp synthetic assemblies, k = 51) from the ORACLE (the C++
# restatement of the reference CPU path, oracle/): writes the 96 FASTA files with the committed generator, runs
# `autocycler_oracle compress` on them (about 26 minutes and 16 GB on 8 vCPUs: end repair 13 min on 8 threads, the k-mer graph,
# unitig graph and simplification 12 min on one core) and records the md5 of the GFA, the printed statistics and the stage times
# in tests/golden/configC_k51.json.  The device test tests/test_gpu_fullsize.py::test_config_c_gfa_digest_equals_the_oracle
# compares the digest of the GFA built on the MI355X from the same inputs with it.
#   bash tests/golden/make_configC_golden.sh [WORKDIR]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=${1:-/tmp/cfgC}
mkdir -p $W
make -C $ROOT/oracle > /dev/null
python - <<EOF
import sys
sys.path.insert(0, "$ROOT")
from autocycler_amd import synth
synth.write_fasta_dir(synth.make_assemblies(96, genome=5_000_000, plasmid=100_000, sub=1e-4, indel=1e-5, seed=51_000), "$W/in")
EOF
S=$(date +%s)
$ROOT/oracle/autocycler_oracle compress -i $W/in -a $W/out --kmer 51 -t 8 2> $W/oracle.log
E=$(( $(date +%s) - S ))
python - <<EOF
import hashlib, json, re
log = open("$W/oracle.log").read()
m = re.search(r"times: load\+repair ([\d.]+) kmer_graph ([\d.]+) unitig_graph ([\d.]+) simplify ([\d.]+) save ([\d.]+)", log)
st = re.findall(r"(\d+) unitigs, (\d+) links\ntotal length: (\d+) bp", log)
md5 = hashlib.md5(open("$W/out/input_assemblies.gfa", "rb").read()).hexdigest()
json.dump({"what": "autocycler_oracle compress (oracle/: C++ restatement of the reference CPU path) on BASELINE configs[2]",
           "inputs": "synth.make_assemblies(96, genome=5_000_000, plasmid=100_000, sub=1e-4, indel=1e-5, seed=51_000) written as FASTA",
           "k": 51, "gfa_md5": md5, "gfa_bytes": len(open("$W/out/input_assemblies.gfa", "rb").read()),
           "kmers": int(re.search(r"Graph contains (\d+) k-mers", log).group(1)),
           "pre": dict(zip(("unitigs", "links", "total_length"), map(int, st[0]))),
           "post": dict(zip(("unitigs", "links", "total_length"), map(int, st[1]))),
           "seconds": dict(zip(("load_and_end_repair_8_threads", "kmer_graph", "unitig_graph", "simplify", "save"), map(float, m.groups()))),
           "wall_seconds": $E}, open("$ROOT/tests/golden/configC_k51.json", "w"), indent=1)
print(md5)
EOF
