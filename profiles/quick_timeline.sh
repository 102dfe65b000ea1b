#!/bin/bash
# One profiled bench run -> the timed region's step in dispatch order (measurement tooling; prints to stdout):
#   gpurun -- 'bash profiles/quick_timeline.sh [bench args]'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
rocprofv3 --kernel-trace -d /tmp/prof_q -o q -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs "$@" > /dev/null 2>&1
python $R/profiles/timeline_rocpd.py /tmp/prof_q/q_results.db 8
