#!/bin/bash
# round 6, call q: A/B of the grouped launch at the Yelp and the stress shape (alternating processes on one box)
mkdir -p gpurun_out
: > gpurun_out/r06q_ab.txt
for i in 1 2 3; do
  for v in 0 1; do
    for wl in yelp stress; do
    echo "$wl pair=$v $(LVAE_PAIR_WGRAD=$v python bench.py --workload $wl --steps 30 --warmup 5 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d.get('roofline_secondary',{}); q=d.get('roofline',{})
print(d['value'], d['ms_per_step'], 'gemm', r.get('ms_per_step'), r.get('achieved'), 'lstm', q.get('ms_per_step'), q.get('us_per_timestep'))")" >> gpurun_out/r06q_ab.txt
    done
  done
done
cat gpurun_out/r06q_ab.txt
