#!/bin/bash
# round 6, session 4, call k: the operand-image conversions in their 16-byte form (cvt_b16_v4_kernel): tests, microbenchmark against the narrow form, default line
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "cvt or dropout_folded or embed_gather_into or gate_weight or gemm_h16 or gemm_b16" > $O/r07k_pytest.txt 2>&1; tail -3 $O/r07k_pytest.txt
timeout 300 python profiles/microbench/cvt_width_ab.py > $O/r07k_cvt_width_ab.txt 2>&1; cat $O/r07k_cvt_width_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/r07k_pytest_parity.txt 2>&1; tail -3 $O/r07k_pytest_parity.txt
timeout 600 python bench.py --steps 40 --warmup 10 --no-side-runs --no-cpu-baseline --no-vendor-baseline > $O/r07k_bench.json 2> $O/r07k_bench.err; cut -c1-250 $O/r07k_bench.json
