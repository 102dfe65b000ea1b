#!/bin/bash
# round 6, session 4, call j: soaks of the persistent recurrences with the 16-byte hand-off granules (varying shapes; plain at B <= 32 and B <= 128,
# with the exchange forced on over one-rank RCCL, the two-pass exact forward) -- ladder rung 0 and no replayed step at the end of each
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python profiles/microbench/soak_persistent.py 6000 32 > $O/r07j_soak_plain.txt 2>&1; tail -2 $O/r07j_soak_plain.txt | cut -c1-250
timeout 900 python profiles/microbench/soak_persistent.py 3000 128 > $O/r07j_soak_b128.txt 2>&1; tail -2 $O/r07j_soak_b128.txt | cut -c1-250
timeout 600 python profiles/microbench/soak_persistent.py 3000 32 dp > $O/r07j_soak_dp.txt 2>&1; tail -3 $O/r07j_soak_dp.txt | cut -c1-250
timeout 600 python profiles/microbench/soak_persistent.py 1500 128 dp > $O/r07j_soak_dp_b128.txt 2>&1; tail -3 $O/r07j_soak_dp_b128.txt | cut -c1-250
timeout 600 python profiles/microbench/soak_persistent.py 2000 32 exact > $O/r07j_soak_exact.txt 2>&1; tail -2 $O/r07j_soak_exact.txt | cut -c1-250
timeout 600 python bench.py --steps 4000 --warmup 10 --no-cpu-baseline --no-side-runs --no-vendor-baseline > $O/r07j_bench_4000_steps.json 2>/dev/null; cut -c1-200 $O/r07j_bench_4000_steps.json
