#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python -m pytest tests -q -s -m gpu > $O/r06x4_full.txt 2>&1; echo "full suite (-s) rc=$?"; grep -v "^  File" $O/r06x4_full.txt | grep -B25 "Fatal Python" | cut -c1-400 | tail -40
for sh in 0 2 4; do
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -s -m gpu -k "(gemm_b16_pair and PAIR$sh) or test_gpu_parity" > $O/r06x4_c$sh.txt 2>&1; echo "pair shape $sh + test_gpu_parity.py (-s): rc=$?"
grep -v "^  File" $O/r06x4_c$sh.txt | grep -B25 "Fatal Python" | cut -c1-400 | tail -30
done
