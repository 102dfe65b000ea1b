#!/bin/bash
# Step-level A/B of an alternative library build against the product build on ONE box (measurement tooling):
#   gpurun -- 'bash profiles/microbench/ab_step.sh <variant name> [bench args]'   (variant built with build_variant.sh)
V=$1; shift
cp vae_lagging_encoder_amd/csrc/liblvae_hip.so /tmp/prod.so
for i in 1 2 3; do for lib in prod $V; do
if [ $lib = prod ]; then cp /tmp/prod.so vae_lagging_encoder_amd/csrc/liblvae_hip.so; else cp profiles/microbench/liblvae_$lib.so vae_lagging_encoder_amd/csrc/liblvae_hip.so; fi
python bench.py --no-side-runs --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['roofline']['ms_per_step'], d['roofline_secondary']['ms_per_step'], d.get('rest_ms_per_step'))"
done; done
cp /tmp/prod.so vae_lagging_encoder_amd/csrc/liblvae_hip.so
