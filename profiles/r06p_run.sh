#!/bin/bash
# round 6, call p: the whole GPU suite and the default bench line on the tree with the grouped launch
mkdir -p gpurun_out
python -m pytest tests -q -x -m gpu > gpurun_out/r06p_pytest_gpu.txt 2>&1
tail -4 gpurun_out/r06p_pytest_gpu.txt
python bench.py > gpurun_out/r06p_bench_default.json 2> gpurun_out/r06p_bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06p_bench_default.json').readline())
print({k:d.get(k) for k in ['value','ms_per_step','value_200_steps']})
print('roofline', {k:d['roofline'].get(k) for k in ['achieved','frac','ms_per_step','us_per_timestep']})
print('secondary', {k:d['roofline_secondary'].get(k) for k in ['achieved','frac','ms_per_step','launches_per_step']})
for k,v in d.get('side_runs',{}).items():
    print(k, {kk:v.get(kk) for kk in ['value','unit','ms_per_step']} if isinstance(v,dict) else v)
print('dropin', d.get('dropin_path')); print('kl_exact', d.get('kl_exact_path'))
PY
