set -x
mkdir -p gpurun_out/r03e
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03e/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03e/pytest.log
tail -6 gpurun_out/r03e/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03e/bench_default.json 2> gpurun_out/r03e/bench_default.err; cut -c1-250 gpurun_out/r03e/bench_default.json
timeout 300 python bench.py --workload stress --no-cpu-baseline > gpurun_out/r03e/bench_stress.json 2> gpurun_out/r03e/bench_stress.err; cut -c1-250 gpurun_out/r03e/bench_stress.json
timeout 300 python bench.py --workload yelp --no-cpu-baseline --no-side-runs > gpurun_out/r03e/bench_yelp.json 2> gpurun_out/r03e/bench_yelp.err; cut -c1-250 gpurun_out/r03e/bench_yelp.json
timeout 300 python bench.py --graph 1 --no-cpu-baseline --no-side-runs > gpurun_out/r03e/bench_graph.json 2> gpurun_out/r03e/bench_graph.err; cut -c1-250 gpurun_out/r03e/bench_graph.json
