#!/bin/bash
# round 6, session 4, call l: softmax statistics merge with its loads in one round trip, 16-byte gradient scaling in the SGD launch: tests + default line + kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "sgd or clip or softmax or nll or optim or norm" > $O/r07l_pytest.txt 2>&1; tail -3 $O/r07l_pytest.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/r07l_pytest_parity.txt 2>&1; tail -3 $O/r07l_pytest_parity.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_r07l -o r07l -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs --no-vendor-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $O/prof_r07l/r07l_results.db > $O/r07l_kernel_stats.txt; python profiles/timeline_rocpd.py $O/prof_r07l/r07l_results.db 8 > $O/r07l_timeline.txt; rm -rf $O/prof_r07l
grep -i "cvt_b16\|merge\|sgd" $O/r07l_kernel_stats.txt | cut -c1-200
tail -2 $O/r07l_timeline.txt
timeout 600 python bench.py --steps 40 --warmup 10 --no-side-runs --no-cpu-baseline --no-vendor-baseline > $O/r07l_bench.json 2> $O/r07l_bench.err; cut -c1-250 $O/r07l_bench.json
