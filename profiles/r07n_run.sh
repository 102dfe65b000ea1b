#!/bin/bash
# round 6, session 4, call n: embedding-gradient scatter with the slice's rows looked up through a cooperatively fetched window of the sorted list
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "embed or scatter" > $O/r07n_pytest.txt 2>&1; tail -3 $O/r07n_pytest.txt
timeout 300 python profiles/microbench/embed_scatter_zipf.py > $O/r07n_embed_scatter_zipf.txt 2>&1; tail -6 $O/r07n_embed_scatter_zipf.txt | cut -c1-250
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/r07n_pytest_parity.txt 2>&1; tail -3 $O/r07n_pytest_parity.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_r07n -o r07n -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs --no-vendor-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $O/prof_r07n/r07n_results.db > $O/r07n_kernel_stats.txt; rm -rf $O/prof_r07n
grep -i "scatter" $O/r07n_kernel_stats.txt | cut -c1-200
timeout 600 python bench.py --tokens zipf --steps 40 --warmup 10 --no-side-runs --no-cpu-baseline --no-vendor-baseline > $O/r07n_bench_zipf.json 2> $O/r07n_bench_zipf.err; cut -c1-200 $O/r07n_bench_zipf.json
