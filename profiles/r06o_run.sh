#!/bin/bash
# round 6, call o: the hand-off with a closing contributor (bounded look for the others, own piece stays in registers)
mkdir -p gpurun_out
M=profiles/microbench
LVAE_PROBE_LIBS=$M/liblvae_skabl1.so,$M/liblvae_skspin0.so python $M/gemm_pair_probe.py > gpurun_out/r06o_pair_probe.txt 2>&1
python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "pair" > gpurun_out/r06o_pytest_pair.txt 2>&1
cat gpurun_out/r06o_pair_probe.txt; tail -3 gpurun_out/r06o_pytest_pair.txt
