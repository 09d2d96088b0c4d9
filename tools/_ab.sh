mkdir -p gpurun_out
V="base;AC_TABLE_HALF=1;base"
AC_NO_TORCH=1 timeout 300 python tools/ab_knobs.py --workload configEprime_k51 --steps 4 --variants "$V" > gpurun_out/r06l_ab_E.txt 2>&1
AC_NO_TORCH=1 timeout 600 python tools/ab_knobs.py --workload configD_k101 --steps 3 --variants "$V" > gpurun_out/r06l_ab_D.txt 2>&1
AC_NO_TORCH=1 timeout 600 python tools/ab_knobs.py --workload configEmini_k51 --steps 3 --variants "$V" > gpurun_out/r06l_ab_Em.txt 2>&1
python - <<'PY'
import json
for f in ["E", "D", "Em"]:
    for line in open(f"gpurun_out/r06l_ab_{f}.txt"):
        try: d = json.loads(line)
        except Exception: print(line[:300]); continue
        if "variant" in d:
            st = d.get("stages_ms", {})
            print(f, d["variant"], round(d["ms_median"], 3), "cap", d.get("table_capacity"), "ins", st.get("insert"), "coll", st.get("collect_sort"), "deg", st.get("degree"), "paths", st.get("paths"), "links", st.get("links"), d.get("gfa_md5"), d.get("error"))
PY
