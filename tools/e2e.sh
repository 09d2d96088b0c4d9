#!/bin/bash
# Whole-command timing (T_e2e of SURVEY.md §8d) on the GPU box: synthetic config-C FASTA directory -> autocycler-compress CLI
# -> input_assemblies.gfa / .yaml, cold and warm.  Usage: tools/e2e.sh TAG [N=96]
TAG=${1:-rXX}; N=${2:-96}
mkdir -p gpurun_out
D=/tmp/e2e_in; O=/tmp/e2e_out
python - <<PY
import sys, time
sys.path.insert(0, ".")
from autocycler_amd import synth
t=time.time(); synth.write_fasta_dir(synth.make_assemblies($N, seed=51000), "$D"); print("fasta written in %.2fs" % (time.time()-t))
PY
du -sh $D | cut -f1
for run in 1 2; do
  rm -rf $O; ./autocycler_amd/autocycler-compress -i $D -a $O -t 32 2>&1 | grep -E "Stage times|Time to run|unitigs" | tr '\n' ' '; echo
done
ls -la $O; md5sum $O/input_assemblies.gfa
rm -rf $O; ./autocycler_amd/autocycler-compress -i $D -a $O -t 32 2>/dev/null; md5sum $O/input_assemblies.gfa
