# round 6, session c: (1) per-timestep phase split of the persistent recurrences from the LV_TRACE build of the FINAL kernels
# (VERDICT r5 item 4); (2) kernel trace of the data-parallel step's compute-side overhead on one GPU (bench.py --force-dp: a
# one-rank RCCL group runs every collective and every wire conversion of the schedule) next to the plain step
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
bash profiles/microbench/build_trace.sh > $O/build_trace.txt 2>&1
tail -2 $O/build_trace.txt
python profiles/microbench/lstm_trace_probe.py > $O/lstm_phase_split.txt 2>&1
cat $O/lstm_phase_split.txt
python profiles/microbench/lstm_fixed_cost_probe.py > $O/lstm_fixed_cost.txt 2>&1
tail -12 $O/lstm_fixed_cost.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_dp -o dp -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-dp --steps 20 --warmup 5 --no-cpu-baseline --no-side-runs > $GRAFT_REPO_ROOT/$O/bench_force_dp.json 2> $GRAFT_REPO_ROOT/$O/bench_force_dp.err
cd $GRAFT_REPO_ROOT
ls -R $O/prof_dp | head -20
f=$(find $O/prof_dp -name "*kernel_stats.csv" | head -1)
python - "$f" > $O/force_dp_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --kernel-trace --stats of `bench.py --gpus 1 --force-dp --steps 20 --warmup 5` (one-rank RCCL group)")
for r in rows[:45]:
    print("%-110s calls %5s total_ms %9.3f avg_us %9.2f pct %5s" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
head -50 $O/force_dp_kernel_stats.txt
rm -rf $O/prof_dp
