export TMPDIR=/tmp
R=$PWD
AB_VARIANTS="base;base" bash tools/gpu_visit.sh r14e tests ab ab:configEprime_k51 ab:configEmini_k51 ab:configDprime_k201
cd /tmp && AC_NO_TORCH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r14e_profE -o stats -- python $R/tools/ab_knobs.py --workload configEprime_k51 --steps 4 --variants "base" > $R/gpurun_out/r14e_prof_ab_E.json 2> $R/gpurun_out/r14e_profE.err; cd $R
DB=$(find gpurun_out/r14e_profE -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB > gpurun_out/r14e_kernel_stats_driver_configEprime.csv && head -45 gpurun_out/r14e_kernel_stats_driver_configEprime.csv | cut -c1-150
find gpurun_out/r14e_profE -type f -size +8M -delete
timeout 1200 python tools/fullsize_e_time.py --builds 2 > gpurun_out/r14e_fullsize_e_time.json 2> gpurun_out/r14e_fullsize.err; tail -c 1200 gpurun_out/r14e_fullsize_e_time.json
