# round 5, call j: dW_ih and dW_hh of each LSTM as ONE product with two destinations (lv_gemm_b16_dual): kernel test, parity, same-box A/B
mkdir -p gpurun_out/r05j
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "dual" > gpurun_out/r05j/pytest.txt 2>&1; tail -3 gpurun_out/r05j/pytest.txt
for d in 1 0 1 0 1 0; do LVAE_DUAL_WGRAD=$d timeout 300 python bench.py --steps 30 --warmup 5 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('dual=$d', d['value'], d['ms_per_step'], 'gemm', d['roofline_secondary']['ms_per_step'], d['roofline_secondary']['achieved'], d['roofline_secondary']['launches_per_step'], 'rest', d['rest_ms_per_step'])"; done
for d in 1 0; do LVAE_DUAL_WGRAD=$d timeout 300 python bench.py --workload yelp --steps 30 --warmup 5 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('yelp dual=$d', d['value'], d['ms_per_step'])"; done
