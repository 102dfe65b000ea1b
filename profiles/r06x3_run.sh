#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "(gemm_b16_pair and PAIR4) or test_gpu_parity" > $O/r06x3_plain.txt 2>&1; echo "plain rc=$?"; tail -1 $O/r06x3_plain.txt | cut -c1-200
timeout 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGABRT stop print" -ex run -ex "bt 40" -ex "info threads" --args python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -k "(gemm_b16_pair and PAIR4) or test_gpu_parity" > $O/r06x3_gdb.txt 2>&1; echo "gdb rc=$?"
grep -n "SIGABRT\|^#" $O/r06x3_gdb.txt | head -60 | cut -c1-250
tail -5 $O/r06x3_gdb.txt | cut -c1-250
