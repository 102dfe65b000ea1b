#!/bin/bash
# round 6, session 4, call f: 16-byte reduce-scatter granules in the 8- / 16-row persistent BPTT (LV_RS_Q = 2) against the build before them,
# and the 16-row input block length (LV_SBB16) under the new register budget
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python profiles/microbench/lstm_swap_ab.py profiles/microbench/liblvae_before_bq.so "before (BPTT 8 / 16 rows: 8-byte granules)" > $O/r07f_lstm_bq_ab.txt 2>&1; echo rc=$?; cat $O/r07f_lstm_bq_ab.txt
timeout 900 python profiles/microbench/lstm_swap_ab.py profiles/microbench/liblvae_p16sbb4.so "LV_SBB16=4" > $O/r07f_lstm_bq_sbb4.txt 2>&1; echo rc=$?; cat $O/r07f_lstm_bq_sbb4.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm or persist" > $O/r07f_pytest_lstm.txt 2>&1; tail -5 $O/r07f_pytest_lstm.txt
