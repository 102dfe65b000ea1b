set -x
mkdir -p gpurun_out/r03g
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03g/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03g/pytest.log
tail -4 gpurun_out/r03g/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03g/bench_default.json 2> gpurun_out/r03g/bench_default.err; cut -c1-200 gpurun_out/r03g/bench_default.json
timeout 300 python bench.py --workload stress --no-cpu-baseline > gpurun_out/r03g/bench_stress.json 2> gpurun_out/r03g/bench_stress.err; cut -c1-200 gpurun_out/r03g/bench_stress.json
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03g/prof -o r03g -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-side-runs > $GRAFT_REPO_ROOT/gpurun_out/r03g/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03g/prof.err; ls $GRAFT_REPO_ROOT/gpurun_out/r03g/prof | head
