mkdir -p gpurun_out/r04g
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "gemm_b16 or throughput or headline or trajectory or fused_trainer or timed_out or inner_loop or yahoo or yelp or stress or properties" > gpurun_out/r04g/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04g/pytest.log
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-side-runs > gpurun_out/r04g/bench$i.json 2> gpurun_out/r04g/bench.err; python - <<PY
import json
d=json.load(open('gpurun_out/r04g/bench$i.json'))
print(d['value'], d['ms_per_step'], 'lstm', d['roofline']['ms_per_step'], 'gemm', d['roofline_secondary']['ms_per_step'], d['roofline_secondary']['achieved'], 'rest', d['rest_ms_per_step'])
PY
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r04g/prof -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs > /dev/null 2>&1
cd $R
DB=$(find gpurun_out/r04g/prof -name "*_results.db" | head -1)
python profiles/summarize_rocpd.py $DB > gpurun_out/r04g/kernel_stats.txt
python profiles/timeline_rocpd.py $DB 2 > gpurun_out/r04g/timeline.txt
rm -rf gpurun_out/r04g/prof
tail -3 gpurun_out/r04g/timeline.txt
