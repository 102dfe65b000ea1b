mkdir -p gpurun_out/r04d
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r04d/prof -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs > /dev/null 2>&1
cd $R
DB=$(find gpurun_out/r04d/prof -name "*_results.db" | head -1)
python profiles/summarize_rocpd.py $DB > gpurun_out/r04d/kernel_stats.txt
python profiles/timeline_rocpd.py $DB 2 > gpurun_out/r04d/timeline.txt
rm -rf gpurun_out/r04d/prof
head -40 gpurun_out/r04d/kernel_stats.txt | cut -c1-200
