export TMPDIR=/tmp
bash tools/gpu_visit.sh r14p smoke tests pmc pmc:configEprime_k51 pmc:configEmini_k51 pmc:configD_k101 bench bench:configEprime_k51 bench:configEmini_k51 bench:configD_k101 bench:configDprime_k201 prof
AB_VARIANTS="base;base" bash tools/gpu_visit.sh r14p ab ab:configEprime_k51 ab:configEmini_k51 ab:configD_k101 ab:configDprime_k201
timeout 1200 python tools/fullsize_e_time.py --builds 3 > gpurun_out/r14p_fullsize_e_time.json 2> gpurun_out/r14p_fullsize.err; tail -c 1300 gpurun_out/r14p_fullsize_e_time.json
