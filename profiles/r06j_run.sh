# round 6, session j: price of a 16-bit Gx image on the KL (kl_ablation.py, two new rows); the stress test's measured KL errors; the L2's
# memory-side read requests split by destination (the only view of "Infinity Cache or HBM" this rocprofv3 has)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
python profiles/microbench/kl_ablation.py > $O/kl_ablation.txt 2>&1; grep -E "^==|binary16|bf16 configuration" $O/kl_ablation.txt | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "stress_config_at_full_size" 2>&1 | grep -E "stress B=128|passed|failed" > $O/stress_kl.txt; cat $O/stress_kl.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -s -k "subnormal" 2>&1 | grep -E "binary16 subnormal|passed|failed" > $O/subnormal.txt; cat $O/subnormal.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_ea -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/pmc_ea -name "*counter_collection.csv" | head -1)
python profiles/summarize_ea_split.py $f > $O/ea_read_split.txt 2>&1; head -16 $O/ea_read_split.txt | cut -c1-190
rm -rf $O/pmc_ea
