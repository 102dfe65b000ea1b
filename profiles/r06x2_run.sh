#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
AMD_LOG_LEVEL=2 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "(gemm_b16_pair and PAIR4) or test_gpu_parity" > $O/r06x2_log.txt 2>&1; echo "rc=$?"
grep -n "Fatal Python" $O/r06x2_log.txt | head -3
grep -v "^  File" $O/r06x2_log.txt | grep -B40 "Fatal Python" | cut -c1-300 | tail -60
