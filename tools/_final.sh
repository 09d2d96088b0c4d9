mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r07b_gpu_pytest.log 2>&1; tail -2 gpurun_out/r07b_gpu_pytest.log
bash tools/gpu_final.sh r07b
bash tools/gpu_timeline.sh r07b_tl > /dev/null; tail -1 gpurun_out/r07b_tl_timeline.txt
AC_NO_TORCH=1 timeout 300 python tools/ab_knobs.py --workload configE2_k51 --steps 3 --variants "base" > gpurun_out/r07b_ab_E2.txt 2>&1
AC_NO_TORCH=1 timeout 300 python tools/ab_knobs.py --workload configEmini_k51 --steps 3 --variants "base;base" > gpurun_out/r07b_ab_Em.txt 2>&1
AC_NO_TORCH=1 timeout 300 python tools/ab_knobs.py --workload configB_k51 --steps 5 --variants "base" > gpurun_out/r07b_ab_B.txt 2>&1
python - <<'PY'
import json
for f in ["E2", "Em", "B"]:
    for line in open(f"gpurun_out/r07b_ab_{f}.txt"):
        try: d = json.loads(line)
        except Exception: print(line[:300]); continue
        if "variant" in d: print(f, d["variant"], round(d["ms_median"], 3), d.get("unitigs"), d.get("stages_ms"), d.get("gfa_md5"), d.get("error"))
PY
