#!/bin/bash
# round 6, session 3, call b: what would 16-byte partial-sum granules buy the persistent BPTT?  What-if builds (LV_P16_ABL bit 6)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python profiles/microbench/lstm_anatomy_probe.py 0 2 64 66 > $O/r07b_lstm_anatomy_gran16.txt 2>&1; echo rc=$?; cat $O/r07b_lstm_anatomy_gran16.txt
