#!/bin/bash
# round 6, session 4, call h: anatomy of a timestep of the persistent recurrences WITH the 16-byte hand-off granules (what-if builds)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python profiles/microbench/lstm_anatomy_probe.py 1 2 3 8 15 > $O/r07h_lstm_anatomy_q.txt 2>&1; echo rc=$?; cat $O/r07h_lstm_anatomy_q.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "lstm or persist" > $O/r07h_pytest_lstm.txt 2>&1; tail -3 $O/r07h_pytest_lstm.txt
