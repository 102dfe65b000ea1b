#!/bin/bash
# round 6, call r: the closing piece of an aligned tile made longer (LV_SK_SKEW K tiles; product build 3) against the even split
mkdir -p gpurun_out
M=profiles/microbench
LVAE_PROBE_LIBS=$M/liblvae_skew0.so,$M/liblvae_skew2.so,$M/liblvae_skew5.so python $M/gemm_pair_probe.py > gpurun_out/r06r_pair_probe.txt 2>&1
python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "pair" > gpurun_out/r06r_pytest_pair.txt 2>&1
grep -v "max|\|vs f64" gpurun_out/r06r_pair_probe.txt; tail -3 gpurun_out/r06r_pytest_pair.txt
