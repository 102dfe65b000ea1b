#!/bin/bash
# round 6, call s: kernel trace of the headline bench on the tree with the grouped launch (one step in dispatch order)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof_r06s -o r06s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $O/prof_r06s/r06s_results.db > $O/r06s_kernel_stats.txt
python $R/profiles/timeline_rocpd.py $O/prof_r06s/r06s_results.db 8 > $O/r06s_timeline.txt
rm -rf $O/prof_r06s
cut -c1-170 $O/r06s_timeline.txt
