set -x
mkdir -p gpurun_out/r04a
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "tile256" > gpurun_out/r04a/pytest_tile256.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04a/pytest_tile256.log
M=profiles/microbench
LVAE_PROBE_LIBS=$M/liblvae_ppdma1.so,$M/liblvae_ppdma2.so,$M/liblvae_ppnoprio.so timeout 500 python $M/gemm_pp_probe.py > gpurun_out/r04a/gemm_pp_probe.txt 2>&1; cat gpurun_out/r04a/gemm_pp_probe.txt
timeout 200 python bench.py --no-cpu-baseline --no-side-runs > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; cut -c1-300 gpurun_out/r04a/bench.json
