run() { echo "== $1"; shift; AC_NO_TORCH=1 "$@" python tools/ab_knobs.py --host-entry --steps 12 --variants 'base;AC_UPLOAD_THREADS=48;AC_UPLOAD_THREADS=64;base' 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    if 'variant' in j: print('  ', j['variant'], '| ms', round(j['ms_median'],3), 'upload', round(j.get('upload_device_ms',0),3))
"; }
run default env
run "cpus 0-63" taskset -c 0-63
run "cpus 64-127" taskset -c 64-127
run "cpus 0-63,128-191 (socket 0 with SMT)" taskset -c 0-63,128-191
run "every 2nd core both sockets" taskset -c 0,2,4,6,8,10,12,14,16,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64,66,68,70,72,74,76,78,80,82,84,86,88,90,92,94,96,98,100,102,104,106,108,110,112,114,116,118,120,122,124,126
