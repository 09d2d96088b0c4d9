export TMPDIR=/tmp; R=$PWD; TAG=$1; WL=$2
cd /tmp && AC_NO_TORCH=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$WL -o stats -- python $R/tools/ab_knobs.py --workload $WL --steps 3 --variants "base" > $R/gpurun_out/${TAG}_prof_ab_$WL.json 2> $R/gpurun_out/${TAG}_prof_$WL.err; cd $R
DB=$(find gpurun_out/${TAG}_prof_$WL -name '*.db' | head -1)
python tools/rocpd_stats.py $DB > gpurun_out/${TAG}_kernel_stats_driver_$WL.csv
python tools/rocpd_launches.py $DB "" 5000 0 > gpurun_out/${TAG}_timeline_driver_$WL.txt 2>/dev/null
find gpurun_out/${TAG}_prof_$WL -type f -size +8M -delete
head -45 gpurun_out/${TAG}_kernel_stats_driver_$WL.csv
