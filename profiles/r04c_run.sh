mkdir -p gpurun_out/r04c
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04c/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04c/pytest_gpu.log
