#!/bin/bash
# Pins the oracle AND the product on the real reference binary, the day a Rust toolchain is at hand (none in the build image:
# SURVEY.md 8c — `cargo`, `rustc` absent, no network; until then whole-GFA bytes are pinned by restatement + the reference's own
# round-trip properties only, and the YAML side output is "parity unpinned").
#
#   tests/golden/check_against_rust.sh [REFERENCE_DIR=/root/reference] [WORK_DIR=/tmp/ac_rust_check]
#
# If `cargo` exists: builds REFERENCE_DIR (cargo build --release --offline), runs `autocycler compress` on
#   (1) the reference's five fixed sequences as the five kinds of assembly file (tests.rs:131-148) at k = 13 and k = 51,
#   (2) BASELINE configs[1] (synth.WORKLOADS["configB_k51"]: 12 x ~5 Mbp, k = 51) unless AC_RUST_CHECK_SMALL=1,
# and compares input_assemblies.gfa AND input_assemblies.yaml byte for byte with
#   (a) the oracle (oracle/autocycler_oracle: the C++ restatement) and
#   (b) the product CLI (autocycler_amd/autocycler-compress; needs the MI355X — skipped with a note when no GPU is visible).
# Exit 0 = identical or nothing to compare with (no cargo: prints why and exits 0, so that CI without Rust stays green); 1 = a difference.
set -u
REF=${1:-/root/reference}
WORK=${2:-/tmp/ac_rust_check}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
if ! command -v cargo >/dev/null 2>&1; then
  echo "check_against_rust: no cargo on PATH - the reference cannot be built here; nothing compared (oracle parity stays 'restatement + KATs')."
  exit 0
fi
if [ ! -f "$REF/Cargo.toml" ]; then echo "check_against_rust: $REF/Cargo.toml not found; nothing compared."; exit 0; fi
mkdir -p "$WORK" || exit 1
( cd "$REF" && CARGO_TARGET_DIR="$WORK/target" cargo build --release --offline ) || { echo "check_against_rust: cargo build failed (offline build needs the vendored crates)"; exit 0; }
BIN="$WORK/target/release/autocycler"
[ -x "$BIN" ] || { echo "check_against_rust: $BIN missing after the build"; exit 1; }
make -C "$ROOT/oracle" >/dev/null || exit 1
ORACLE="$ROOT/oracle/autocycler_oracle"
PRODUCT="$ROOT/autocycler_amd/autocycler-compress"
HAVE_GPU=0
python3 - <<PY && HAVE_GPU=1
import sys; sys.path.insert(0, "$ROOT")
from autocycler_amd import _capi
import os; os.environ["AC_NO_TORCH"] = "1"
sys.exit(0 if _capi.load_library().ac_device_count() >= 1 else 1)
PY
python3 - "$WORK" <<PY || exit 1
import sys; sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
from pathlib import Path
import os
import boundary_cases as B
from autocycler_amd import synth
work = Path(sys.argv[1])
B.write_five_file_fixture(work / "five")
if not os.environ.get("AC_RUST_CHECK_SMALL"):
    k, n, gen = synth.WORKLOADS["configB_k51"]
    synth.write_fasta_dir(gen(), work / "configB")
PY
rc=0
compare() {   # name, input dir, k
  local name=$1 in=$2 k=$3
  rm -rf "$WORK/out_rust_$name" "$WORK/out_oracle_$name" "$WORK/out_product_$name"
  "$BIN" compress -i "$in" -a "$WORK/out_rust_$name" --kmer "$k" >/dev/null 2>"$WORK/rust_$name.log" || { echo "$name: the reference binary failed (see $WORK/rust_$name.log)"; rc=1; return; }
  "$ORACLE" compress -i "$in" -a "$WORK/out_oracle_$name" --kmer "$k" >/dev/null 2>&1 || { echo "$name: the oracle failed"; rc=1; return; }
  for f in input_assemblies.gfa input_assemblies.yaml; do
    if cmp -s "$WORK/out_rust_$name/$f" "$WORK/out_oracle_$name/$f"; then echo "$name k=$k $f: oracle == reference"; else echo "$name k=$k $f: ORACLE DIFFERS FROM THE REFERENCE"; rc=1; fi
  done
  if [ $HAVE_GPU = 1 ] && [ -x "$PRODUCT" ]; then
    "$PRODUCT" compress -i "$in" -a "$WORK/out_product_$name" --kmer "$k" >/dev/null 2>&1 || { echo "$name: the product CLI failed"; rc=1; return; }
    for f in input_assemblies.gfa input_assemblies.yaml; do
      if cmp -s "$WORK/out_rust_$name/$f" "$WORK/out_product_$name/$f"; then echo "$name k=$k $f: product == reference"; else echo "$name k=$k $f: PRODUCT DIFFERS FROM THE REFERENCE"; rc=1; fi
    done
  else
    echo "$name: no MI355X visible (or the CLI is not built) - product not compared"
  fi
}
compare five13 "$WORK/five" 13
compare five51 "$WORK/five" 51
[ -d "$WORK/configB" ] && compare configB "$WORK/configB" 51
exit $rc
