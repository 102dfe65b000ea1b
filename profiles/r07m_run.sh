#!/bin/bash
# round 6, session 4, call m (run twice: staging first, then the LDS dot products batched): the batch-sized head / tail kernels with their staging loads in flight together (stage_batched) and the dz partial sums as 16-byte loads
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "head or tail or dec_init or enc_" > $O/r07m_pytest.txt 2>&1; tail -3 $O/r07m_pytest.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/r07m_pytest_parity.txt 2>&1; tail -3 $O/r07m_pytest_parity.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_r07m -o r07m -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs --no-vendor-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $O/prof_r07m/r07m_results.db > $O/r07m_kernel_stats.txt; python profiles/timeline_rocpd.py $O/prof_r07m/r07m_results.db 8 > $O/r07m_timeline.txt; rm -rf $O/prof_r07m
grep -i "head\|tail_bwd\|dec_init" $O/r07m_kernel_stats.txt | cut -c1-200
tail -1 $O/r07m_timeline.txt
timeout 600 python bench.py --steps 40 --warmup 10 --no-side-runs --no-cpu-baseline --no-vendor-baseline > $O/r07m_bench.json 2> $O/r07m_bench.err; cut -c1-250 $O/r07m_bench.json
