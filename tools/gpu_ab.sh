#!/bin/bash
# One short GPU-box visit without torch: A/B of the tuning knobs on config C (same device text, graph digests), then the
# device parity tests with the new kernel variants switched on.  Usage: tools/gpu_ab.sh TAG
TAG=${1:-rXX}
mkdir -p gpurun_out
export AC_NO_TORCH=1
V="base;AC_DEGREE_VARIANT=1;AC_MINKEY_VARIANT=1;AC_TABLE_SHIFT=1;AC_PATH_CHUNK=128;AC_PATH_CHUNK=512;AC_DEGREE_VARIANT=1,AC_MINKEY_VARIANT=1;AC_DEGREE_VARIANT=1,AC_MINKEY_VARIANT=1,AC_TABLE_SHIFT=1;base"
timeout 120 python tools/ab_knobs.py --variants "$V" > gpurun_out/${TAG}_ab.jsonl 2> gpurun_out/${TAG}_ab.err; echo "ab exit $?"
cut -c1-330 gpurun_out/${TAG}_ab.jsonl
tail -3 gpurun_out/${TAG}_ab.err
AC_DEGREE_VARIANT=1 AC_MINKEY_VARIANT=1 timeout 170 python -m pytest tests/test_gpu_parity.py -x -q \
  -k "fixed_seqs or adversarial or key_word or synthetic_assemblies_medium or renumber_tie or many_path" > gpurun_out/${TAG}_parity_variants.log 2>&1
echo "parity exit $?"; tail -3 gpurun_out/${TAG}_parity_variants.log
