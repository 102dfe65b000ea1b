#!/bin/bash
# round 6, session 4, call r: the whole GPU suite twice more on the final tree (flake check) + smoke + the default line
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for i in 1 2; do ( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/r07r_pytest_gpu_$i.txt 2>&1; tail -5 $O/r07r_pytest_gpu_$i.txt | head -2; done
python __graft_entry__.py --smoke > $O/r07r_smoke.txt 2>&1; tail -1 $O/r07r_smoke.txt
python bench.py > $O/r07r_bench_default.json 2> $O/r07r_bench_default.err; cut -c1-200 $O/r07r_bench_default.json
