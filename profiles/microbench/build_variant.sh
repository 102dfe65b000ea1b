#!/bin/bash
# builds profiles/microbench/liblvae_<name>.so = the product sources with extra -D flags, for the A/B columns of the microbenchmarks
# (measurement only, git-ignored).   usage: build_variant.sh <name> -DLV_B16_PP_DMA=1 ...
cd "$(dirname "$0")/../.."
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" \
  -I vae_lagging_encoder_amd/csrc -o profiles/microbench/liblvae_$NAME.so vae_lagging_encoder_amd/csrc/*.hip 2>&1 | grep -v "warning\|note\|^ *[0-9]* |\|^ *|\|\^" | head -20
