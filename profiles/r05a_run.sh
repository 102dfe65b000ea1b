# round 5, call a: the self-launching bench (2 ranks on the one GPU, gloo), the KL ablation, the exact-encoder-forward configuration
mkdir -p gpurun_out/r05a
timeout 600 python -m pytest tests/test_bench_launch.py tests/test_gpu_kernels.py -q -m gpu -k "bench_gpus_2 or import_saved" > gpurun_out/r05a/pytest_new.txt 2>&1; tail -5 gpurun_out/r05a/pytest_new.txt
timeout 600 python profiles/microbench/kl_ablation.py > gpurun_out/r05a/kl_ablation.txt 2>&1; cat gpurun_out/r05a/kl_ablation.txt | grep -v Warning | tail -30
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "exact_encoder_forward or bf16_headline" > gpurun_out/r05a/pytest_exact.txt 2>&1; tail -15 gpurun_out/r05a/pytest_exact.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05a/bench_default.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "kl_exact_path", "f32_parity_path", "elbo_delta_vs_cpu")})
PY
timeout 300 python bench.py --steps 20 --warmup 5 --encoder-forward f32 --no-side-runs --no-cpu-baseline > gpurun_out/r05a/bench_exact_fwd.json 2> gpurun_out/r05a/bench_exact_fwd.err; cut -c1-400 gpurun_out/r05a/bench_exact_fwd.json; tail -3 gpurun_out/r05a/bench_exact_fwd.err
