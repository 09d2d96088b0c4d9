export TMPDIR=/tmp
AB_VARIANTS="base;base" bash tools/gpu_visit.sh r14g ab:configEprime_k51 ab:configEmini_k51
timeout 1200 python tools/fullsize_e_time.py --builds 2 > gpurun_out/r14g_fullsize_e_time.json 2> gpurun_out/r14g_fullsize.err; tail -c 1300 gpurun_out/r14g_fullsize_e_time.json
