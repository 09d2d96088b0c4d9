#!/usr/bin/env python3
"""Golden results from the ORACLE (oracle/: the C++ restatement of the reference CPU path) for a named workload of
autocycler_amd.synth.WORKLOADS: writes the FASTA files with the committed generator, runs `autocycler_oracle compress` on them
and records the md5 of the GFA, the printed statistics and the stage times in tests/golden/NAME.json.
tests/test_gpu_fullsize.py::test_gfa_digest_equals_the_oracle compares the GFA built on the MI355X from the same inputs with it.

    python tests/golden/make_golden.py configEprime_k51        (5 species x 20 strains x ~1 Mbp, k = 51)

(configB/C/Dprime were recorded by make_configC_golden.sh, the shell predecessor of this script; same flow.)"""
import hashlib
import json
import os
import re
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from autocycler_amd import synth

name = sys.argv[1]
k, n_asm, gen = synth.WORKLOADS[name]
work = Path(os.environ.get("GOLDEN_WORKDIR", f"/tmp/golden_{name}"))
work.mkdir(parents=True, exist_ok=True)
subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)
synth.write_fasta_dir(gen(), work / "in")
t0 = time.time()
with open(work / "oracle.log", "w") as log:
    subprocess.check_call([str(ROOT / "oracle" / "autocycler_oracle"), "compress", "-i", str(work / "in"), "-a", str(work / "out"),
                           "--kmer", str(k), "-t", "8"], stderr=log)
wall = time.time() - t0
log = (work / "oracle.log").read_text()
m = re.search(r"times: load\+repair ([\d.]+) kmer_graph ([\d.]+) unitig_graph ([\d.]+) simplify ([\d.]+) save ([\d.]+)", log)
st = re.findall(r"(\d+) unitigs, (\d+) links\ntotal length: (\d+) bp", log)
data = (work / "out" / "input_assemblies.gfa").read_bytes()
peak = re.search(r"peak RSS ([\d.]+) GB", log)
out = {"what": "autocycler_oracle compress (oracle/: C++ restatement of the reference CPU path)",
       "inputs": f"autocycler_amd.synth.WORKLOADS[{name!r}] written as FASTA (synth.write_fasta_dir)",
       "k": k, "assemblies": n_asm, "gfa_md5": hashlib.md5(data).hexdigest(), "gfa_bytes": len(data),
       "kmers": int(re.search(r"Graph contains (\d+) k-mers", log).group(1)),
       "pre": dict(zip(("unitigs", "links", "total_length"), map(int, st[0]))),
       "post": dict(zip(("unitigs", "links", "total_length"), map(int, st[1]))),
       "seconds": dict(zip(("load_and_end_repair_8_threads", "kmer_graph", "unitig_graph", "simplify", "save"), map(float, m.groups()))),
       "wall_seconds": round(wall),
       **({"oracle_env": {"ORACLE_NO_POSITION_RESERVE": "1 (capacity hint of kmer_graph.rs:40 not applied; output unaffected)"}} if os.environ.get("ORACLE_NO_POSITION_RESERVE") else {}),
       "host": "build container: 8 vCPUs, 62 GB; hot stages (k-mer graph, unitig graph, simplify) on one core like the reference"}
(ROOT / "tests" / "golden" / f"{name}.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
