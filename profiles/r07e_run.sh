#!/bin/bash
# round 6, session 4, call e: 16-byte h granules in the 4-row persistent forward (LV_FWD4_Q) against the build before them:
# outputs side by side, microseconds per timestep, then the LSTM kernel tests and the parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python profiles/microbench/lstm_swap_ab.py profiles/microbench/liblvae_before_fq.so "before (forward: 8-byte granules)" > $O/r07e_lstm_fq_ab.txt 2>&1; echo rc=$?; cat $O/r07e_lstm_fq_ab.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm or persist" > $O/r07e_pytest_lstm.txt 2>&1; tail -5 $O/r07e_pytest_lstm.txt


