#!/bin/bash
# Regenerates the judged measurement set of a round on the GPU box (through gpurun):
#   bash profiles/run_round_profiles.sh r02p
# writes gpurun_out/<tag>_*; copy what is to be judged into profiles/.  PMC passes are separate runs with --kernel-trace only.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
python bench.py --graph 1 --no-cpu-baseline --no-side-runs > $O/${TAG}_bench_hipgraph.json 2>/dev/null
python bench.py --workload yelp --no-cpu-baseline --no-side-runs > $O/${TAG}_bench_yelp.json 2>/dev/null
python bench.py --workload stress --no-cpu-baseline --no-side-runs > $O/${TAG}_bench_stress.json 2>/dev/null
python bench.py --dtype f32 --no-cpu-baseline --no-side-runs > $O/${TAG}_bench_yahoo_f32.json 2>/dev/null
# round 5: the split-bf16 / two-pass exact encoder forward as its own line, the bf16 forward operands of rounds 1-4 as the A/B of the
# binary16 default, and the self-launching data-parallel entry (2 ranks; on a one-GPU box they share the device and exchange over gloo)
python bench.py --encoder-forward f32 --no-cpu-baseline --no-side-runs > $O/${TAG}_bench_kl_exact.json 2>/dev/null
python bench.py --forward-operands bf16 --no-cpu-baseline --no-side-runs > $O/${TAG}_bench_bf16_forward_operands.json 2>/dev/null
python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_gpus2_selflaunch.json 2> $O/${TAG}_bench_gpus2_selflaunch.err
# round 6: the other SCALE points on the shared device (host-staged gloo), the driver's torchrun form, and the exchange on a one-rank RCCL group
python bench.py --gpus 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_gpus4_selflaunch.json 2> $O/${TAG}_bench_gpus4_selflaunch.err
python bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_gpus8_selflaunch.json 2> $O/${TAG}_bench_gpus8_selflaunch.err
LVAE_DIST_BACKEND=gloo LVAE_SHARED_GPU="2 ranks on 1 GPU (torchrun form)" python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29733 \
  bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_gpus2_torchrun.json 2> $O/${TAG}_bench_gpus2_torchrun.err
python bench.py --gpus 1 --force-dp --steps 30 --warmup 5 --no-cpu-baseline --no-side-runs > $O/${TAG}_bench_force_dp_rccl_one_rank.json 2> $O/${TAG}_bench_force_dp_rccl_one_rank.err
# Omniglot (BASELINE.json configs[3]): the dtype is named explicitly (the decoder's convolutions are exact f32 in both modes;
# --dtype only moves the encoder's im2col GEMMs) so that file names and the "dtype" field cannot disagree
python bench.py --workload omniglot --dtype f32 --steps 30 --warmup 5 > $O/${TAG}_bench_omniglot_f32.json 2>/dev/null
python bench.py --workload omniglot --dtype f32 --graph 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_omniglot_f32_hipgraph.json 2>/dev/null
python bench.py --workload omniglot --dtype bf16 --graph 1 --steps 30 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_omniglot_bf16_hipgraph.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof_${TAG} -o ${TAG} -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $O/prof_${TAG}/${TAG}_results.db > $O/${TAG}_bench_yahoo_bf16_kernel_stats.txt
# one step of the TIMED region in dispatch order (the last 5 steps of the process are the bench's untimed GEMM-event pass)
python $R/profiles/timeline_rocpd.py $O/prof_${TAG}/${TAG}_results.db 8 > $O/${TAG}_bench_yahoo_bf16_timeline.txt
rocprofv3 --kernel-trace -d $O/prof_omni_${TAG} -o ${TAG}o -- python $R/bench.py --workload omniglot --dtype f32 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py $O/prof_omni_${TAG}/${TAG}o_results.db > $O/${TAG}_omniglot_kernel_stats.txt
for W in yahoo yelp; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${C}_${W}_${TAG} -o p -- \
      python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs > /dev/null 2>&1
  done
  python $R/profiles/summarize_pmc.py $O/pmc_FETCH_SIZE_${W}_${TAG}/p_counter_collection.csv $O/pmc_WRITE_SIZE_${W}_${TAG}/p_counter_collection.csv \
      > $O/${TAG}_pmc_hbm_traffic_${W}_bf16.txt
  python $R/profiles/summarize_pmc.py $O/pmc_FETCH_SIZE_${W}_${TAG}/p_counter_collection.csv $O/pmc_WRITE_SIZE_${W}_${TAG}/p_counter_collection.csv \
      --json 5 > $O/${TAG}_pmc_groups_${W}.json
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq_${TAG} -o p -- \
  python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs > /dev/null 2>&1
python $R/profiles/summarize_sq.py $O/pmc_sq_${TAG}/p_counter_collection.csv > $O/${TAG}_pmc_sq_mfma_busy_yahoo_bf16.txt 2>&1
cd $R
timeout 200 python profiles/microbench/lstm_persist16_probe.py > $O/${TAG}_persist16_probe.txt 2>&1
timeout 100 python profiles/microbench/lstm_fixed_cost_probe.py > $O/${TAG}_lstm_fixed_cost.txt 2>&1
rm -rf $O/prof_${TAG} $O/prof_omni_${TAG} $O/pmc_*_${TAG}
cut -c1-400 $O/${TAG}_bench_default.json
cut -c1-200 $O/${TAG}_bench_hipgraph.json $O/${TAG}_bench_yelp.json $O/${TAG}_bench_stress.json $O/${TAG}_bench_yahoo_f32.json
cut -c1-200 $O/${TAG}_bench_kl_exact.json $O/${TAG}_bench_bf16_forward_operands.json $O/${TAG}_bench_gpus2_selflaunch.json; tail -2 $O/${TAG}_bench_gpus2_selflaunch.err
cut -c1-160 $O/${TAG}_bench_gpus4_selflaunch.json $O/${TAG}_bench_gpus8_selflaunch.json $O/${TAG}_bench_gpus2_torchrun.json $O/${TAG}_bench_force_dp_rccl_one_rank.json; wc -l $O/${TAG}_bench_force_dp_rccl_one_rank.json
cut -c1-200 $O/${TAG}_bench_omniglot_f32.json $O/${TAG}_bench_omniglot_f32_hipgraph.json $O/${TAG}_bench_omniglot_bf16_hipgraph.json
head -8 $O/${TAG}_omniglot_kernel_stats.txt | cut -c1-170
head -14 $O/${TAG}_bench_yahoo_bf16_kernel_stats.txt | cut -c1-170
head -8 $O/${TAG}_pmc_hbm_traffic_yahoo_bf16.txt | cut -c1-170
cat $O/${TAG}_pmc_groups_yahoo.json | tr -d '\n' | cut -c1-600; echo
head -8 $O/${TAG}_pmc_sq_mfma_busy_yahoo_bf16.txt | cut -c1-170
