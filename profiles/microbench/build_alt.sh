#!/bin/bash
# builds liblvae_alt.so = the product sources with extra -D flags (default: the register-staged NT GEMM, LV_B16_GLDS=0) for
# the A/B column of gemm_b16_shapes.py (measurement only, git-ignored).   usage: build_alt.sh [-DLV_B16_GLDS=2 ...]
cd "$(dirname "$0")/../.."
FLAGS="${@:--DLV_B16_GLDS=0}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $FLAGS \
  -I vae_lagging_encoder_amd/csrc -o profiles/microbench/liblvae_alt.so vae_lagging_encoder_amd/csrc/*.hip
