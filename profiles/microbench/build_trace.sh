#!/bin/bash
# Measurement tooling: a SEPARATE build of the kernel library with the phase-trace marks of the persistent LSTM kernels compiled in
# (-DLV_TRACE); profiles/microbench/lstm_trace_probe.py loads it.  The product library has no trace code.
set -e
HERE=$(cd $(dirname $0) && pwd)
CSRC=$HERE/../../vae_lagging_encoder_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DLV_TRACE $LV_TRACE_DEFS -I $CSRC -o $HERE/liblvae_trace.so $CSRC/*.hip
echo built $HERE/liblvae_trace.so
