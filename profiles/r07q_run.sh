#!/bin/bash
# round 6, session 4, call q: a sleep in front of the FIRST poll of a timestep (LV_PREPOLL_F / _B = 1, 2, 3, 4, 6 x 64 cycles): does the phase of the polls matter?
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for v in pp1 pp2 pp3 pp4 pp6; do
timeout 600 python profiles/microbench/lstm_swap_ab.py profiles/microbench/liblvae_p16$v.so "$v" > $O/r07q_prepoll_$v.txt 2>&1; echo rc=$?
done
grep -h "^B=" $O/r07q_prepoll_pp*.txt | sed 's/bit-identical outputs (T=40 \/ T=200): //' | cut -c1-200
