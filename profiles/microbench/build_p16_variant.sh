#!/bin/bash
# builds profiles/microbench/liblvae_p16<name>.so = the product library with lv_lstm_persist16.hip alone recompiled under extra -D flags
# (the other objects come from csrc/build/: run `python -m vae_lagging_encoder_amd.build` first).  Measurement only, git-ignored.
#   usage: build_p16_variant.sh abl66 -DLV_P16_ABL=66
cd "$(dirname "$0")/../.."
NAME=$1; shift
C=vae_lagging_encoder_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -I $C -c $C/lv_lstm_persist16.hip -o /tmp/p16_$NAME.o 2>&1 | grep -E "error" | head
OBJS=$(ls $C/build/*.o | grep -v lv_lstm_persist16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/microbench/liblvae_p16$NAME.so $OBJS /tmp/p16_$NAME.o && echo built liblvae_p16$NAME.so
