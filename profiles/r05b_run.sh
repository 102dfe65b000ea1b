# round 5, call b: the whole GPU suite on the new tree; default bench line (kl_exact_path on the split / two-pass forward, dropin_path,
# mixed_shapes, value_with_token_sort, CPU legs); the exact-forward configuration on its own
mkdir -p gpurun_out/r05b
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r05b/pytest_gpu.txt 2>&1; tail -8 gpurun_out/r05b/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05b/bench_default.json 2> gpurun_out/r05b/bench_default.err; tail -3 gpurun_out/r05b/bench_default.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05b/bench_default.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "mean_loss_per_seq", "host_reads_in_timed_region", "value_with_token_sort", "kl_exact_path", "f32_parity_path", "dropin_path", "cpu_baseline", "rest_ms_per_step")})
print("elbo", d.get("elbo_delta_vs_cpu", {}).get("per_dtype"))
for k, v in d.get("side_runs", {}).items():
    print(k, {a: b for a, b in v.items() if a not in ("workload", "dominant_group", "other_groups")} if isinstance(v, dict) else v)
PY
timeout 300 python bench.py --steps 20 --warmup 5 --encoder-forward f32 --no-side-runs --no-cpu-baseline > gpurun_out/r05b/bench_exact_fwd.json 2> gpurun_out/r05b/bench_exact_fwd.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05b/bench_exact_fwd.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "rest_ms_per_step")}, d["roofline"]["per_recurrence"], d["roofline_secondary"]["ms_per_step"])
PY
