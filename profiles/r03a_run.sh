set -x
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
timeout 300 python profiles/microbench/lstm_persist16_probe.py > gpurun_out/r03a/persist16_probe.txt 2>&1; tail -20 gpurun_out/r03a/persist16_probe.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03a/bench_default.json 2> gpurun_out/r03a/bench_default.err; cut -c1-600 gpurun_out/r03a/bench_default.json
timeout 300 python bench.py --workload stress --no-cpu-baseline > gpurun_out/r03a/bench_stress.json 2> gpurun_out/r03a/bench_stress.err; cut -c1-400 gpurun_out/r03a/bench_stress.json
timeout 300 python bench.py --workload yelp --no-cpu-baseline --no-side-runs > gpurun_out/r03a/bench_yelp.json 2> gpurun_out/r03a/bench_yelp.err; cut -c1-300 gpurun_out/r03a/bench_yelp.json
LVAE_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --persistent 0 --no-cpu-baseline > gpurun_out/r03a/bench_dp2_gloo.json 2> gpurun_out/r03a/bench_dp2_gloo.err; cut -c1-300 gpurun_out/r03a/bench_dp2_gloo.json; tail -3 gpurun_out/r03a/bench_dp2_gloo.err
