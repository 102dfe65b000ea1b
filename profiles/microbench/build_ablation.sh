#!/bin/bash
# builds liblvae_abl<N>.so (ablation variants of lv_gemm_b16; measurement only, git-ignored) next to this script
cd "$(dirname "$0")/../.."
for n in 1 2 3 4 6 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DLV_B16_ABL=$n \
    -I vae_lagging_encoder_amd/csrc -o profiles/microbench/liblvae_abl$n.so vae_lagging_encoder_amd/csrc/*.hip 2>/dev/null &
done
wait
