# round 6, session k: soak of the persistent launches over 3000 steps of varying shape -- plain, and with the data-parallel exchange forced
# on over a one-rank RCCL group (RCCL's stream beside the persistent launches: the liveness assumption of DESIGN section 6)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
timeout 600 python profiles/microbench/soak_persistent.py 3000 32 > $O/soak_plain.txt 2>&1; tail -2 $O/soak_plain.txt | cut -c1-250
timeout 600 python profiles/microbench/soak_persistent.py 3000 32 dp > $O/soak_dp.txt 2>&1; tail -3 $O/soak_dp.txt | cut -c1-250
timeout 600 python profiles/microbench/soak_persistent.py 1500 128 dp > $O/soak_dp_b128.txt 2>&1; tail -3 $O/soak_dp_b128.txt | cut -c1-250
