#!/bin/bash
# round 6, session 4, call o: a short sleep after a FAILED poll of a hand-off (LV_POLL_SLEEP = 1 / 2 / 4 x 64 cycles): less polling traffic in the L2 against later detection
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for v in sl1 sl2 sl4; do
timeout 600 python profiles/microbench/lstm_swap_ab.py profiles/microbench/liblvae_p16$v.so "$v" > $O/r07o_poll_sleep_$v.txt 2>&1; echo rc=$?
done
grep -h "^B=" $O/r07o_poll_sleep_sl*.txt | sed 's/bit-identical outputs (T=40 \/ T=200): //' | cut -c1-200
