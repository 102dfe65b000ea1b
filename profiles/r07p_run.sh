#!/bin/bash
# round 6, session 4, call p: long soaks of the final kernels (the 2-bit / 16-bit tags of the 16-byte granules over many launches and shapes)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1500 python profiles/microbench/soak_persistent.py 40000 32 > $O/r07p_soak_plain.txt 2>&1; tail -2 $O/r07p_soak_plain.txt | cut -c1-250
timeout 1500 python profiles/microbench/soak_persistent.py 20000 128 > $O/r07p_soak_b128.txt 2>&1; tail -2 $O/r07p_soak_b128.txt | cut -c1-250
timeout 900 python profiles/microbench/soak_persistent.py 10000 64 dp > $O/r07p_soak_dp_b64.txt 2>&1; tail -2 $O/r07p_soak_dp_b64.txt | cut -c1-250
timeout 900 python bench.py --steps 20000 --warmup 10 --no-cpu-baseline --no-side-runs --no-vendor-baseline > $O/r07p_bench_20000_steps.json 2>/dev/null; cut -c1-200 $O/r07p_bench_20000_steps.json
timeout 900 python bench.py --workload stress --steps 3000 --warmup 10 --no-cpu-baseline --no-side-runs > $O/r07p_bench_stress_3000_steps.json 2>/dev/null; cut -c1-200 $O/r07p_bench_stress_3000_steps.json
