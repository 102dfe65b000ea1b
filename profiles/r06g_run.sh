# round 6, session g: Omniglot BatchNorm calls against their floors; the 16-row BPTT with four receive rounds (LV_HB16=1, now
# really four); kernel trace of the data-parallel step's compute side (--force-dp on one GPU)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
python profiles/microbench/omniglot_bn_floor.py bf16x3 > $O/omniglot_bn_floor_bf16x3.txt 2>&1
cut -c1-170 $O/omniglot_bn_floor_bf16x3.txt
python profiles/microbench/omniglot_bn_floor.py f32 > $O/omniglot_bn_floor_f32.txt 2>&1
tail -3 $O/omniglot_bn_floor_f32.txt | cut -c1-200
python profiles/microbench/lstm_anatomy_probe.py 0 hb1 > $O/persist16_hb1.txt 2>&1
cut -c1-130 $O/persist16_hb1.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_dp -o dp -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-dp --steps 20 --warmup 5 --no-cpu-baseline --no-side-runs > $GRAFT_REPO_ROOT/$O/bench_force_dp.json 2> $GRAFT_REPO_ROOT/$O/bench_force_dp.err
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $O/prof_dp/dp_results.db > $O/force_dp_kernel_stats.txt 2>&1
python profiles/timeline_rocpd.py $O/prof_dp/dp_results.db 8 > $O/force_dp_timeline.txt 2>&1
head -42 $O/force_dp_kernel_stats.txt | cut -c1-100,112-175
rm -rf $O/prof_dp
