set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
rocm-smi --showid 2>&1 | head -5 > gpurun_out/r06a/smi.txt
python -c "import torch;print(torch.cuda.device_count())" >> gpurun_out/r06a/smi.txt 2>&1
timeout 900 python -m pytest tests/test_rccl_single_rank.py -x -q -m gpu > gpurun_out/r06a/rccl_single.txt 2>&1
tail -30 gpurun_out/r06a/rccl_single.txt
timeout 600 python bench.py --gpus 1 --force-dp --steps 30 --warmup 5 --no-cpu-baseline --no-side-runs > gpurun_out/r06a/bench_force_dp.json 2> gpurun_out/r06a/bench_force_dp.err
tail -c 1500 gpurun_out/r06a/bench_force_dp.json; tail -5 gpurun_out/r06a/bench_force_dp.err
timeout 400 python bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline --launch-timeout 200 > gpurun_out/r06a/bench_gpus4.json 2> gpurun_out/r06a/bench_gpus4.err
echo rc=$?; tail -c 600 gpurun_out/r06a/bench_gpus4.json; tail -60 gpurun_out/r06a/bench_gpus4.err
timeout 600 python -m pytest tests/test_bench_launch.py -x -q -m gpu > gpurun_out/r06a/bench_launch.txt 2>&1
tail -5 gpurun_out/r06a/bench_launch.txt
