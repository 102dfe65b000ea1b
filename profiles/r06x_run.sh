#!/bin/bash
# round 6, call x: narrowing the abort (pair kernel test + engine pair launches in one process)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "gemm_b16_pair or test_gpu_parity" > $O/r06x_a.txt 2>&1; echo "pair kernel test + test_gpu_parity.py: rc=$?"; tail -1 $O/r06x_a.txt | cut -c1-200
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "gemm_b16_pair or image" > $O/r06x_b.txt 2>&1; echo "pair kernel test + image tests: rc=$?"; tail -1 $O/r06x_b.txt | cut -c1-200
for sh in 0 1 2 3 4; do
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "(gemm_b16_pair and PAIR$sh) or test_gpu_parity" > $O/r06x_c$sh.txt 2>&1; echo "pair shape $sh + test_gpu_parity.py: rc=$?"; tail -1 $O/r06x_c$sh.txt | cut -c1-200
done
