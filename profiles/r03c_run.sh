set -x
mkdir -p gpurun_out/r03c
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "persistent16 or throughput_path_h1024 or stress or adam_optim" > gpurun_out/r03c/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03c/pytest.log
tail -4 gpurun_out/r03c/pytest.log
timeout 300 python profiles/microbench/lstm_persist16_probe.py > gpurun_out/r03c/persist16_probe.txt 2>&1; tail -14 gpurun_out/r03c/persist16_probe.txt
timeout 300 python bench.py --workload stress --no-cpu-baseline > gpurun_out/r03c/bench_stress.json 2> gpurun_out/r03c/bench_stress.err; cut -c1-300 gpurun_out/r03c/bench_stress.json
