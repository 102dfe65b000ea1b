set -x
mkdir -p gpurun_out/r03f
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "persistent16 or throughput_path_h1024 or stress or headline or trajectory or inner_loop or properties" > gpurun_out/r03f/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03f/pytest.log
tail -4 gpurun_out/r03f/pytest.log
timeout 300 python profiles/microbench/lstm_persist16_probe.py > gpurun_out/r03f/persist16_probe.txt 2>&1; head -4 gpurun_out/r03f/persist16_probe.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-runs > gpurun_out/r03f/bench_default.json 2> gpurun_out/r03f/bench_default.err; cut -c1-200 gpurun_out/r03f/bench_default.json
