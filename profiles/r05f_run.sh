# round 5, call f: Omniglot with the convolutions' weight-gradient launches on a side stream (graph branch): parity, A/B under hipGraph
mkdir -p gpurun_out/r05f
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "image" > gpurun_out/r05f/pytest_image.txt 2>&1; tail -4 gpurun_out/r05f/pytest_image.txt
for sw in 0 1 0 1; do for dt in f32 bf16x3; do
LVAE_SIDE_WGRADS=$sw timeout 300 python bench.py --workload omniglot --dtype $dt --graph 1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('side_wgrads=$sw', '$dt', d['value'], d['ms_per_step'])"
done; done
LVAE_SIDE_WGRADS=1 timeout 300 python bench.py --workload omniglot --dtype bf16x3 --graph 0 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-160
LVAE_SIDE_WGRADS=0 timeout 300 python bench.py --workload omniglot --dtype bf16x3 --graph 0 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-160
