#!/bin/bash
# round 6, call n: what the grouped launch's 117 us are made of -- what-if builds (no hand-off / no C stores) and other row splits
mkdir -p gpurun_out
M=profiles/microbench
LVAE_PROBE_LIBS=$M/liblvae_skabl1.so,$M/liblvae_skabl3.so,$M/liblvae_skr23.so,$M/liblvae_skr25.so,$M/liblvae_skr26.so python $M/gemm_pair_probe.py > gpurun_out/r06n_pair_probe.txt 2>&1
cat gpurun_out/r06n_pair_probe.txt
