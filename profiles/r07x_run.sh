# round 6, last session, the FINAL tree (the scatter lookup in too): the whole GPU suite + smoke on the final tree, then the judged measurement set (profiles/run_round_profiles.sh r07x)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/r07x_pytest_gpu.txt 2>&1; tail -6 $O/r07x_pytest_gpu.txt
python __graft_entry__.py --smoke > $O/r07x_smoke.txt 2>&1; tail -2 $O/r07x_smoke.txt
bash profiles/run_round_profiles.sh r07x > $O/r07x_profiles_log.txt 2>&1; tail -60 $O/r07x_profiles_log.txt | cut -c1-260
