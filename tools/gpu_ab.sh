#!/bin/bash
# One short GPU-box visit without torch: A/B of the tuning knobs on config C (same device text, graph digests), the device
# parity tests, and a rocprofv3 kernel trace of a few builds (per-launch timeline of one build).  Usage: tools/gpu_ab.sh TAG "VARIANTS" [parity-env]
TAG=${1:-rXX}
V=${2:-"base;AC_TABLE_SHIFT=0,AC_MINKEY_VARIANT=0;base"}
mkdir -p gpurun_out
export AC_NO_TORCH=1 TMPDIR=/tmp
R=$PWD
timeout 120 python tools/ab_knobs.py --variants "$V" > gpurun_out/${TAG}_ab.jsonl 2> gpurun_out/${TAG}_ab.err; echo "ab exit $?"
cut -c1-250 gpurun_out/${TAG}_ab.jsonl
tail -3 gpurun_out/${TAG}_ab.err
env $3 timeout 170 python -m pytest tests/test_gpu_parity.py -x -q \
  -k "fixed_seqs or adversarial or key_word or synthetic_assemblies_medium or renumber_tie or many_path or wide_keys or pairwise or high_diversity" > gpurun_out/${TAG}_parity.log 2>&1
echo "parity exit $?"; tail -3 gpurun_out/${TAG}_parity.log
cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o prof -- python $R/tools/ab_knobs.py --variants base --steps 4 > $R/gpurun_out/${TAG}_ab_under_rocprof.jsonl 2> $R/gpurun_out/${TAG}_rocprof.err; echo "rocprof exit $?"
cd $R
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_kernel_stats.csv; head -12 gpurun_out/${TAG}_kernel_stats.csv
  python tools/rocpd_launches.py $DB "" 400 2000 > gpurun_out/${TAG}_timeline.txt     # launches 2001..2400: inside a steady-state build
  python tools/rocpd_launches.py $DB insert_wave 16 32 > gpurun_out/${TAG}_insert_launches.txt; cat gpurun_out/${TAG}_insert_launches.txt
  rm -f $DB
fi
