#!/bin/bash
# round 6, session 3, call c: the persistent recurrences with DPP / lane-swap instructions in place of ds_bpermute shuffles -- bit identity
# against the build before the change, microseconds per timestep, then the kernel tests
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python profiles/microbench/lstm_swap_ab.py profiles/microbench/liblvae_before_swap.so "before (ds_bpermute shuffles)" > $O/r07c_lstm_swap_ab.txt 2>&1; echo rc=$?; cat $O/r07c_lstm_swap_ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "lstm or persist" > $O/r07c_pytest_lstm.txt 2>&1; tail -3 $O/r07c_pytest_lstm.txt
