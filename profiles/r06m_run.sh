#!/bin/bash
# round 6, call m: grouped stream-K launch of the LSTM-sized gradient products (lv_gemm_b16_pair) -- probe, kernel tests, A/B in the step
mkdir -p gpurun_out
python profiles/microbench/gemm_pair_probe.py > gpurun_out/r06m_pair_probe.txt 2>&1
python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "pair or dual or gemm_b16" > gpurun_out/r06m_pytest_gemm.txt 2>&1
for i in 1 2 3; do
  for v in 0 1; do
    echo "pair=$v $(LVAE_PAIR_WGRAD=$v python bench.py --steps 30 --warmup 5 --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d.get('roofline_secondary',{})
print(d['value'], d['ms_per_step'], 'gemm', r.get('ms_per_step'), r.get('achieved'), r.get('launches_per_step'))")" >> gpurun_out/r06m_ab.txt
  done
done
cat gpurun_out/r06m_pair_probe.txt; tail -5 gpurun_out/r06m_pytest_gemm.txt; cat gpurun_out/r06m_ab.txt
