#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for i in 1 2 3; do
python -m pytest tests -x -q -m gpu > $O/r06x5_full_$i.txt 2>&1; echo "full suite run $i rc=$?"; tail -1 $O/r06x5_full_$i.txt | cut -c1-200
done
for sh in 0 1 2 3 4; do
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "(gemm_b16_pair and PAIR$sh) or test_gpu_parity" > $O/r06x5_c$sh.txt 2>&1; echo "pair shape $sh + test_gpu_parity.py: rc=$?"
done
