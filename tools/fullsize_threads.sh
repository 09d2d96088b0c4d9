# configs[4] at full size: where its d2h stage goes, by host thread count / device share of the stretch renumbering (AC_DEBUG_ARENA prints the split)
export AC_DEBUG_ARENA=1
for V in ${VARIANTS:-"AC_UPLOAD_THREADS=24"}; do
  env $(echo $V | tr ',' ' ') timeout 600 python tools/fullsize_e_time.py --builds 3 > gpurun_out/${TAG:-r15g}_fullsize_$V.json 2> gpurun_out/${TAG:-r15g}_fullsize_$V.err
  echo "== $V"; grep "^d2h" gpurun_out/${TAG:-r15g}_fullsize_$V.err | tail -4 | cut -c1-200; python -c "
import json; j=json.loads(open('gpurun_out/${TAG:-r15g}_fullsize_$V.json').read().strip().splitlines()[-1]); print(j['build_s'], 'd2h', j['stages_ms_last_build']['d2h'], 'finalize', j['stages_ms_last_build']['finalize'], 'verify_failed', j.get('verify_failed'))"
done
