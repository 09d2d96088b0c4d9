#!/bin/bash
# Launch-by-launch timeline of the LAST build of a torch-free driver run (rocprofv3 --kernel-trace; tools/rocpd_launches.py).
# Usage: tools/gpu_timeline.sh TAG [ab_knobs args...]   -> gpurun_out/TAG_timeline.txt, gpurun_out/TAG_kernel_stats.csv
# (TIMELINE_MARKER = the kernel whose last launch begins the last build: PackFunctor for the device entry, MaskTableFunctor for --host-entry)
TAG=$1; shift
export TMPDIR=/tmp AC_NO_TORCH=1
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o prof -- python $R/tools/ab_knobs.py --variants base --steps 1 "$@" > $R/gpurun_out/${TAG}_run.jsonl 2> $R/gpurun_out/${TAG}_rocprof.err
echo "rocprof exit $?"
cd $R
DB=$(find gpurun_out/${TAG}_prof -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_kernel_stats.csv
  python - $DB ${TIMELINE_MARKER:-PackFunctor} > gpurun_out/${TAG}_timeline.txt <<'PY'
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# the last build = everything after the last PackFunctor launch
last = max(i for i, r in enumerate(rows) if sys.argv[2] in r[0])
rows = rows[last:]
t0 = rows[0][1]; prev = t0; busy = 0
for i, (name, s, e) in enumerate(rows):
    short = re.sub(r"^void ", "", name).replace("ac::", "")
    short = re.sub(r"rocprim::ROCPRIM_\w+::detail::", "rocprim::", short)
    print(f"{i:4d} t={(s - t0) / 1e3:9.1f} us  dur={(e - s) / 1e3:8.1f}  gap={(s - prev) / 1e3:7.1f}  {short[:90]}")
    busy += e - s; prev = max(prev, e)
print(f"# {len(rows)} launches, busy {busy / 1e3:.1f} us, span {(prev - t0) / 1e3:.1f} us")
PY
  rm -f $DB
  tail -1 gpurun_out/${TAG}_timeline.txt
fi
