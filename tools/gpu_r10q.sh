#!/bin/bash
# visit q: host-side path renumbering at configs[4] size (82 M unitigs, 1.2 G path entries) and at 8 species
export TMPDIR=/tmp
mkdir -p gpurun_out
for SP in 8 25; do for HR in 1 0; do
  AC_HOST_REMAP=$HR timeout 600 python tools/config_e.py --species $SP --builds 3 > gpurun_out/r10q_configE_${SP}species_hostremap$HR.json 2> gpurun_out/r10q_${SP}_$HR.log; echo "species $SP host_remap $HR exit $?"
  python - gpurun_out/r10q_configE_${SP}species_hostremap$HR.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(d["build_s"], d["stages_ms"])
PY
done; done
