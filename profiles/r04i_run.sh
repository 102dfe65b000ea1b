mkdir -p gpurun_out/r04i
for i in 1 2 3; do timeout 200 python bench.py --no-cpu-baseline --no-side-runs > gpurun_out/r04i/bench$i.json 2> gpurun_out/r04i/bench.err; python - <<PY
import json
d=json.load(open('gpurun_out/r04i/bench$i.json'))
print(d['value'], d['ms_per_step'], 'lstm', d['roofline']['ms_per_step'], d['roofline']['us_per_timestep'], 'gemm', d['roofline_secondary']['ms_per_step'], d['roofline_secondary']['achieved'], 'rest', d['rest_ms_per_step'])
PY
done
timeout 200 python bench.py --workload stress --no-cpu-baseline --no-side-runs > gpurun_out/r04i/stress.json 2>> gpurun_out/r04i/bench.err; cut -c1-120 gpurun_out/r04i/stress.json
tail -2 gpurun_out/r04i/bench.err
