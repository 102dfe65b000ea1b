set -x
mkdir -p gpurun_out/r03i
python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "pixelcnn or batchnorm_eval or image" > gpurun_out/r03i/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03i/pytest.log
tail -12 gpurun_out/r03i/pytest.log
timeout 300 python bench.py --workload omniglot --dtype f32 --steps 30 --warmup 5 > gpurun_out/r03i/bench_omni_f32.json 2> gpurun_out/r03i/bench_omni.err; cut -c1-1500 gpurun_out/r03i/bench_omni_f32.json
