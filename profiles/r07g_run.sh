#!/bin/bash
# round 6, session 4, call g: the whole GPU suite + smoke on the tree with the 16-byte hand-off granules, then the default line and the side workloads
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/r07g_pytest_gpu.txt 2>&1; tail -6 $O/r07g_pytest_gpu.txt
python __graft_entry__.py --smoke > $O/r07g_smoke.txt 2>&1; tail -2 $O/r07g_smoke.txt
python bench.py --no-cpu-baseline --no-vendor-baseline > $O/r07g_bench_default.json 2> $O/r07g_bench_default.err; cut -c1-300 $O/r07g_bench_default.json
python bench.py --workload stress --no-cpu-baseline --no-side-runs > $O/r07g_bench_stress.json 2>/dev/null; cut -c1-300 $O/r07g_bench_stress.json
python bench.py --workload yelp --no-cpu-baseline --no-side-runs > $O/r07g_bench_yelp.json 2>/dev/null; cut -c1-300 $O/r07g_bench_yelp.json
