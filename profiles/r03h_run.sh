set -x
mkdir -p gpurun_out/r03h
python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "throughput_path_h1024 or token_sort or inner_loop or trajectory" > gpurun_out/r03h/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03h/pytest.log
tail -3 gpurun_out/r03h/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-runs > gpurun_out/r03h/bench_default.json 2> gpurun_out/r03h/bench_default.err; cut -c1-200 gpurun_out/r03h/bench_default.json
timeout 300 python bench.py --workload omniglot --dtype f32 --steps 30 --warmup 5 > gpurun_out/r03h/bench_omni_f32.json 2> gpurun_out/r03h/bench_omni.err; cut -c1-300 gpurun_out/r03h/bench_omni_f32.json
timeout 300 python bench.py --workload omniglot --dtype f32 --graph 1 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03h/bench_omni_f32_graph.json 2>> gpurun_out/r03h/bench_omni.err; cut -c1-300 gpurun_out/r03h/bench_omni_f32_graph.json
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03h/prof_omni -o omni -- python $GRAFT_REPO_ROOT/bench.py --workload omniglot --dtype f32 --steps 12 --warmup 3 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r03h/prof_omni.err
