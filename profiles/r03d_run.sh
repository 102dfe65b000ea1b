set -x
mkdir -p gpurun_out/r03d
python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "persistent16" > gpurun_out/r03d/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03d/pytest.log
tail -4 gpurun_out/r03d/pytest.log
timeout 300 python profiles/microbench/lstm_persist16_probe.py > gpurun_out/r03d/persist16_probe.txt 2>&1; tail -14 gpurun_out/r03d/persist16_probe.txt
