# round 5, call e: binary16 forward operands in the encoder (the bf16 configuration's default from here on): kernel tests, parity at the
# headline / Yelp shapes, the ablation table with the new row, bench default (A/B --forward-operands bf16 on the same box)
mkdir -p gpurun_out/r05e
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "h16 or binary16 or persistent" > gpurun_out/r05e/pytest_kernels.txt 2>&1; tail -3 gpurun_out/r05e/pytest_kernels.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16 or exact_encoder or stress_config" > gpurun_out/r05e/pytest_parity.txt 2>&1; tail -12 gpurun_out/r05e/pytest_parity.txt
timeout 600 python profiles/microbench/kl_ablation.py > gpurun_out/r05e/kl_ablation.txt 2>&1; grep "==\|binary16\|bf16 configuration\|exact forward (" gpurun_out/r05e/kl_ablation.txt
for ops in f16 bf16 f16 bf16; do timeout 300 python bench.py --steps 30 --warmup 5 --no-side-runs --no-cpu-baseline --forward-operands $ops 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$ops', d['value'], d['ms_per_step'], d['roofline']['per_recurrence']['fwd_enc'], d['roofline_secondary']['ms_per_step'])"; done
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05e/bench_default.json 2> gpurun_out/r05e/bench_default.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05e/bench_default.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "parity_contract")})
print("elbo", d.get("elbo_delta_vs_cpu", {}).get("per_dtype"))
PY
