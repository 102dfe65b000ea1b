# round 5, call c: persistent launches without memsets (alternating exchange halves) -- kernel tests, fixed-cost probe in both forms,
# parity at the headline shape, bench default + hipGraph line
mkdir -p gpurun_out/r05c
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -x -k "persistent or alternate or headline or exact_encoder or transaction or timed_out or stress_config or fold" > gpurun_out/r05c/pytest_persist.txt 2>&1; tail -4 gpurun_out/r05c/pytest_persist.txt
timeout 300 python profiles/microbench/lstm_fixed_cost_probe.py > gpurun_out/r05c/lstm_fixed_cost.txt 2>&1; grep -v Warning gpurun_out/r05c/lstm_fixed_cost.txt | grep "==\|fixed\|T=200\|T=  1\|status"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05c/bench_default.json 2> gpurun_out/r05c/bench_default.err; tail -2 gpurun_out/r05c/bench_default.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05c/bench_default.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "rest_ms_per_step", "value_with_token_sort")}, d["roofline"]["ms_per_step"], d["roofline"]["us_per_timestep"], d["roofline_secondary"]["ms_per_step"])
print("kl_exact", d.get("kl_exact_path"), "dropin", d.get("dropin_path"))
print("mixed", d["side_runs"].get("mixed_shapes"))
PY
timeout 300 python bench.py --steps 20 --warmup 5 --graph 1 --no-side-runs --no-cpu-baseline > gpurun_out/r05c/bench_hipgraph.json 2> gpurun_out/r05c/bench_hipgraph.err; cut -c1-200 gpurun_out/r05c/bench_hipgraph.json
