#!/bin/bash
# round 6, session 4, call i: cache-policy bits of the XCD-local 16-byte granule store (LV_XCD_ST_MODS): nt / sc0 / sc0 nt against none
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for v in stnt stsc0 stsc0nt; do
timeout 600 python profiles/microbench/lstm_swap_ab.py profiles/microbench/liblvae_p16$v.so "$v" > $O/r07i_store_mods_$v.txt 2>&1; echo rc=$?; grep -v amdgpu.ids $O/r07i_store_mods_$v.txt | cut -c1-20,60-400
done
