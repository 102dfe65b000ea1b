#!/bin/bash
# round 6, session 4, call s: the input projections on the 256 x 256 tile against the shipped 128 x 128 kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python profiles/microbench/gx_tile_probe.py > $O/r07s_gx_tile_probe.txt 2>&1; cat $O/r07s_gx_tile_probe.txt | cut -c1-330
