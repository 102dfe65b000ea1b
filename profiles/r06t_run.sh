#!/bin/bash
# round 6, call t: are the two input projections (Gx: 105 MB of f32 output each) bound by their stores?  what-if build without the epilogue's stores
mkdir -p gpurun_out
LVAE_PROBE_LIBS=profiles/microbench/liblvae_epiabl1.so python profiles/microbench/gemm_lstm_shapes.py > gpurun_out/r06t_gemm_lstm_shapes.txt 2>&1
cut -c1-250 gpurun_out/r06t_gemm_lstm_shapes.txt
