mkdir -p gpurun_out/r04b
M=profiles/microbench
L=""; for v in 6 70 134 198; do L=$L,$M/liblvae_ppabl$v.so; done
LVAE_PROBE_SHAPES=sq8k,dO,logits LVAE_PROBE_TILES=257 LVAE_PROBE_LIBS=${L#,} timeout 500 python $M/gemm_pp_probe.py > gpurun_out/r04b/gemm_pp_ablation3.txt 2>&1; cat gpurun_out/r04b/gemm_pp_ablation3.txt
