#!/bin/bash
# builds profiles/microbench/liblvae_<name>.so = the product library with lv_gemm_b16.hip recompiled under extra -D flags (the other
# objects are the product's own, csrc/build/*.o); measurement only, git-ignored.   usage: build_gemm_variant.sh <name> -DLV_SK_ABL=1 ...
cd "$(dirname "$0")/../.."
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -I vae_lagging_encoder_amd/csrc \
  -c vae_lagging_encoder_amd/csrc/lv_gemm_b16.hip -o /tmp/lv_gemm_b16_$NAME.o 2>&1 | grep -E "error" | head
OBJS=$(ls vae_lagging_encoder_amd/csrc/build/*.o | grep -v lv_gemm_b16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/microbench/liblvae_$NAME.so $OBJS /tmp/lv_gemm_b16_$NAME.o
