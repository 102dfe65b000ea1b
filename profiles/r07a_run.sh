#!/bin/bash
# round 6, session 3, call a: does hipExtAnyOrderLaunch overlap consecutive kernels of one stream on gfx950?
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/any_order_probe profiles/microbench/any_order_probe.hip 2>/dev/null
timeout 120 /tmp/any_order_probe > $O/r07a_any_order_probe.txt 2>&1; echo rc=$?; cat $O/r07a_any_order_probe.txt
