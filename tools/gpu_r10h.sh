#!/bin/bash
# Round-4 visit h: after the wave kernels got one source (wave_rt.hpp) — the whole device suite, then the builds' times against visit f.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r10h_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r10h_pytest_gpu.log
export AC_NO_TORCH=1
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    if "variant" in j:
        st = j.get("stages_ms", {})
        print(j["variant"], "| ms", round(j.get("ms_median", 0), 3), "min", round(j.get("ms_min", 0), 3), "| insert_k", round(j.get("insert_kernel_ms", 0), 3), {k: st.get(k) for k in ("collect_sort", "degree", "minkey", "paths", "expand", "seqs", "finalize", "d2h")}, j.get("gfa_md5", "")[:8], j.get("error", ""))
PY
}
for W in configEprime_k51 configDprime_k101 configEmini_k51; do
  timeout 400 python tools/ab_knobs.py --workload $W --steps 6 --variants "base;base" > gpurun_out/r10h_ab_$W.jsonl 2> gpurun_out/r10h_ab_$W.err; echo "$W exit $?"
  show gpurun_out/r10h_ab_$W.jsonl
done
timeout 300 python tools/ab_knobs.py --steps 8 --variants "base;AC_PATH_COPY=0;base" > gpurun_out/r10h_ab_configC_k51.jsonl 2> gpurun_out/r10h_ab_configC.err; echo "C exit $?"
show gpurun_out/r10h_ab_configC_k51.jsonl
