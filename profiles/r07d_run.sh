#!/bin/bash
# round 6, session 4, call d: 16-byte reduce-scatter granules in the 4-row persistent BPTT (LV_RS4_Q) against the build before them:
# outputs side by side, microseconds per timestep, then the LSTM kernel tests and the parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python profiles/microbench/lstm_swap_ab.py profiles/microbench/liblvae_before_q4.so "before (8-byte granules)" > $O/r07d_lstm_q4_ab.txt 2>&1; echo rc=$?; cat $O/r07d_lstm_q4_ab.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm or persist" > $O/r07d_pytest_lstm.txt 2>&1; tail -5 $O/r07d_pytest_lstm.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/r07d_pytest_parity.txt 2>&1; tail -5 $O/r07d_pytest_parity.txt
timeout 600 python bench.py --steps 40 --warmup 10 --no-side-runs --no-cpu-baseline --no-vendor-baseline > $O/r07d_bench.json 2> $O/r07d_bench.err; tail -c 1500 $O/r07d_bench.json
