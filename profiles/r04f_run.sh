mkdir -p gpurun_out/r04f
M=profiles/microbench
LVAE_PROBE_SHAPES=dO,dW_pred,sq8k LVAE_PROBE_TILES=258 LVAE_PROBE_LIBS=$M/liblvae_qnt.so,$M/liblvae_qsc0.so,$M/liblvae_qsc1.so,$M/liblvae_qsc01.so timeout 500 python $M/gemm_pp_probe.py > gpurun_out/r04f/gemm_q_policy.txt 2>&1; cat gpurun_out/r04f/gemm_q_policy.txt
